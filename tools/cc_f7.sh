# dev helper: compile bf_fp4_stream.hip, print register / spill summary and the hot loop's instruction skeleton
cd /root/repo/zvdb_amd/csrc || exit 1
mkdir -p build /tmp/p
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -Rpass-analysis=kernel-resource-usage -c bf_fp4_stream.hip -o build/bf_fp4_stream.o 2>&1 | grep -A9 "Name: .*k_fp4_streamILi12ELi0ELi32E" | grep "Spill\|Scratch\|VGPRs:\|error"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include -S --cuda-device-only -o /tmp/p/f7.s bf_fp4_stream.hip 2>&1 | grep error
awk '/^_ZN4zvdb12_GLOBAL__N_112k_fp4_streamILi12ELi0ELi32E.*:/,/s_endpgm/' /tmp/p/f7.s > /tmp/p/f7k.s
echo "lines $(wc -l < /tmp/p/f7k.s) mfma $(grep -c v_mfma /tmp/p/f7k.s) scratch $(grep -c scratch_ /tmp/p/f7k.s) s_load $(grep -c 's_load' /tmp/p/f7k.s)"
