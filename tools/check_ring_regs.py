"""ISA audit of the k_f32_regq kernels (bf_f32_regq.hip), every instantiation.

The operand ring lives in registers named LITERALLY inside asm statements, above the compiler's allocation limit (amdgpu_num_vgpr; RA =
rq_acc_base(KS): the group's accumulator, then the ring, then the norm register).  The asm statements are correct only if hipcc never touches those registers and never waits on vmcnt with its own (ring-
blind) count inside the group loop.  This script compiles the TU to ISA and FAILS on, per kernel:
  * any instruction OUTSIDE ;;#ASMSTART / ;;#ASMEND that names a vector register >= RA, or any accumulation register at all;
  * inside the group loop: a compiler-emitted s_waitcnt with a vmcnt field (its vmcnt(0) for a late query fragment would drain the ring once
    per group), a compiler-emitted load or spill (outside the loop a spill is reported, not failed: k_f32_regq_28_256_0 parks 5 registers
    across the whole loop), a ring wait other than vmcnt(R - 1) / the norm's vmcnt(KS - 3), MFMAs != 4 x ring loads,
    ring loads != KS x groups per iteration;
  * a kernel descriptor that does not allocate the whole 256-register file.
History (docs/design/k4_bruteforce.md, K4r): round 6's first ring was inline-asm loads and waits tied to COMPILER-allocated registers; hipcc
copied a ring element AHEAD of its asm wait (v_mov_b64 in most instantiations) and a soak against the int8 route found one lost neighbour
in ~10^5 queries.  Run by tests/test_build_guards.py (CPU: hipcc cross-compiles).     python tools/check_ring_regs.py [file.hip]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ring_len(ks):   # rq_ring_len / rq_unit_groups of bf_f32_regq.hip
    return 16 if ks == 8 else ks if ks <= 20 else {24: 12, 28: 14, 32: 8}[ks]


def unit_groups(ks):
    return 2 if ks == 8 else 1


def vregs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def check(src=None, verbose=True):
    src = src or os.path.join(ROOT, "zvdb_amd", "csrc", "bf_f32_regq.hip")
    asm = "/tmp/check_ring_regs.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "zvdb_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", asm], check=True, stderr=subprocess.DEVNULL)
    whole = open(asm).read()
    text = whole.splitlines()
    bad, seen = [], 0
    i = 0
    while i < len(text):
        m = re.match(r"^(_ZN4zvdb\S*k_f32_regq_(\d+)_(\d+)_(\d)\S*):", text[i])
        if not m:
            i += 1
            continue
        name, ks = m.group(1), int(m.group(2))
        tag = f"k_f32_regq_{m.group(2)}_{m.group(3)}_{m.group(4)}"
        R = ring_len(ks)
        RB = 252 - 4 * R
        RA = RB - 16
        j = i
        while "s_endpgm" not in text[j]:
            j += 1
        body = text[i:j]
        i = j
        seen += 1
        # (1) compiler instructions keep out of the ring's registers
        inasm = False
        kinds = []   # per line: 'asm' / 'cc' / 'mark'
        for l in body:
            if "#ASMSTART" in l:
                inasm = True
                kinds.append("mark")
                continue
            if "#ASMEND" in l:
                inasm = False
                kinds.append("mark")
                continue
            kinds.append("asm" if inasm else "cc")
        for l, kd in zip(body, kinds):
            code = l.split(";")[0].strip()
            if kd != "cc" or not code or code.startswith("."):
                continue
            hi = [r for r in vregs(code) if r >= RA]
            if hi:
                bad.append((tag, f"compiler instruction names v{min(hi)} >= v{RA}, the first register of the asm statements: {code}"))
            if re.search(r"\ba(\[\d|\d)", code) or "accvgpr" in code:
                bad.append((tag, "compiler instruction names an accumulation register: " + code))
        # (2) the group loop
        labels = {l.split(":")[0]: n for n, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        loops = []
        for n, l in enumerate(body):
            mb = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
            if mb and mb.group(1) in labels and labels[mb.group(1)] < n:
                lo, hi2 = labels[mb.group(1)], n
                if sum("v_mfma" in x for x in body[lo:hi2]) >= 32:
                    loops.append((hi2 - lo, lo, hi2))
        if not loops:
            bad.append((tag, "no group loop found"))
            continue
        _, lo, hi2 = min(loops)
        seg = [(l.split(";")[0].strip(), kd) for l, kd in zip(body[lo:hi2], kinds[lo:hi2])]
        seg = [(l, kd) for l, kd in seg if l and not l.startswith(".")]
        nm = sum(l.startswith("v_mfma") for l, _ in seg)
        nl = sum(l.startswith("global_load_dwordx4") for l, _ in seg)
        if nm != 4 * nl or nl != ks * unit_groups(ks):
            bad.append((tag, f"{nm} MFMAs, {nl} ring loads for KS = {ks}"))
        ring_waits = set()
        for l, kd in seg:
            if l.startswith("s_waitcnt") and "vmcnt" in l:
                if kd == "cc":
                    bad.append((tag, "compiler wait on vmcnt inside the group loop: " + l))
                else:
                    ring_waits.add(l)
            if kd == "cc" and l.startswith(("global_load", "flat_load", "buffer_load", "scratch_")):
                bad.append((tag, "compiler load / spill inside the group loop: " + l))
        want = {f"s_waitcnt vmcnt({R - 1})", f"s_waitcnt vmcnt({ks - 3})"}   # the ring's; the norm's
        if not ring_waits <= want or f"s_waitcnt vmcnt({R - 1})" not in ring_waits:
            bad.append((tag, f"ring waits {sorted(ring_waits)}, expected {sorted(want)}"))
        # (3) the descriptor
        md = re.search(r"\.name:\s+" + re.escape(name) + r"\n", whole)
        blk = whole[max(0, md.start() - 1500):md.end() + 800] if md else ""
        get = lambda k: ([int(x) for x in re.findall(k + r":\s+(\d+)", blk)] or [None])[-1]
        vg, sp, pv, ag = get(r"\.vgpr_count"), get(r"\.vgpr_spill_count"), get(r"\.private_segment_fixed_size"), get(r"\.agpr_count")
        if vg != 256 or ag not in (0, None):
            bad.append((tag, f"descriptor: vgpr_count {vg}, agpr_count {ag}, vgpr_spill_count {sp}, private segment {pv}"))
        if verbose:
            top = max([r for l, kd in zip(body, kinds) if kd == "cc" for r in vregs(l.split(";")[0])] or [0])
            print(f"{tag:24s} accumulator v[{RA}:{RA + 15}], ring v[{RB}:{RB + 4 * R - 1}], norms v252; compiler's highest register v{top}; loop: {nm} MFMAs, {nl} ring loads, "
                  f"waits {sorted(ring_waits)}; vgpr_count {vg}, spills {sp}, private segment {pv}")
    for name, what in bad:
        print("FAIL", name, what)
    if verbose:
        print(f"{seen} kernels checked, {len(bad)} findings")
    return seen, bad


if __name__ == "__main__":
    seen, bad = check(sys.argv[1] if len(sys.argv) > 1 else None)
    sys.exit(1 if bad or seen == 0 else 0)
