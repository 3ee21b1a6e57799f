"""ISA guard of the group loop of k_f32_regq (bf_f32_regq.hip), every instantiation.

The operand ring's loads are plain, compiler-visible loads, so correctness is hipcc's own waitcnt pass.  What this script holds is the SHAPE
the kernel's speed depends on (docs/design/k4_bruteforce.md, K4r): inside the group loop
  * every wait on the ring is COUNTED (no s_waitcnt vmcnt(0): with a ring as long as the iteration hipcc carried one element through a copy
    at the loop head behind a full drain, -9 % at configs[1]);
  * no ring register is copied (v_mov_b64 / v_mov_b32 of a load's destination) and nothing is spilled (scratch_*);
  * the loop holds 4 MFMAs per ring load.
History: round 6's first form issued and awaited the ring by inline asm tied to the ring registers; hipcc copied a ring element AHEAD of its
asm wait (two v_mov_b64 in k_f32_regq<8, 128, 0> and others) and a soak against the int8 route found one lost neighbour in ~10^5 queries.
Run by tests/test_build_guards.py (CPU: hipcc cross-compiles).     python tools/check_ring_regs.py [file.hip]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vregs(tok):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def check(src=None, verbose=True):
    src = src or os.path.join(ROOT, "zvdb_amd", "csrc", "bf_f32_regq.hip")
    asm = "/tmp/check_ring_regs.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "zvdb_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", asm], check=True, stderr=subprocess.DEVNULL)
    text = open(asm).read().splitlines()
    bad, seen = [], 0
    i = 0
    while i < len(text):
        m = re.match(r"^(_ZN4zvdb\S*k_f32_regqI\S+):", text[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i
        while "s_endpgm" not in text[j]:
            j += 1
        body = text[i:j]
        i = j
        labels = {l.split(":")[0]: n for n, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        loops = []
        for n, l in enumerate(body):
            mb = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
            if mb and mb.group(1) in labels and labels[mb.group(1)] < n:
                lo, hi = labels[mb.group(1)], n
                seg = body[lo:hi]
                if sum("v_mfma" in x for x in seg) >= 32 and any("global_load_dwordx4" in x for x in seg):
                    loops.append((hi - lo, lo, hi))
        if not loops:
            bad.append((name, "no group loop found"))
            continue
        _, lo, hi = min(loops)   # innermost loop with the MFMAs
        seg = [l.split(";")[0].strip() for l in body[lo:hi]]
        seg = [l for l in seg if l and not l.startswith(".")]
        ring = set()
        for l in seg:
            if l.startswith("global_load_dwordx4"):
                ring |= vregs(l.split(",")[0])
        nm, nl = sum(l.startswith("v_mfma") for l in seg), sum(l.startswith("global_load_dwordx4") for l in seg)
        seen += 1
        if nm != 4 * nl:
            bad.append((name, f"{nm} MFMAs for {nl} ring loads"))
        for l in seg:
            if l.startswith("s_waitcnt") and "vmcnt(0)" in l:
                bad.append((name, "full drain inside the group loop: " + l))
            if l.startswith("scratch_"):
                bad.append((name, "spill inside the group loop: " + l))
            if l.startswith("v_mov_b") and vregs(",".join(l.split(",")[1:])) & ring:
                bad.append((name, "ring register copied: " + l))
        if verbose:
            waits = sorted({l for l in seg if l.startswith("s_waitcnt") and "vmcnt" in l})
            print(f"{name[33:60]:27s} loop of {len(seg)} instructions: {nm} MFMAs, {nl} ring loads, ring = {len(ring)} VGPRs, waits {waits}")
    return seen, bad


if __name__ == "__main__":
    seen, bad = check(sys.argv[1] if len(sys.argv) > 1 else None)
    for name, what in bad:
        print("FINDING:", name[:80], "|", what)
    print(f"{seen} instantiations checked, {len(bad)} findings")
    sys.exit(1 if bad or seen == 0 else 0)
