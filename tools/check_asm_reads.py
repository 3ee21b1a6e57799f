"""ISA audit of the stream kernels' LDS operand rings (bf_bf16_stream.hip by default; also bf_i8_stream.hip, bf_fp4_stream.hip, bruteforce.hip).

Their fragment reads are inline-asm ds_read_b128 on registers hipcc allocates ("=&v"), awaited by separate asm s_waitcnt lgkmcnt(N)
statements tied to the same registers ("+v").  That pins ORDER, not register allocation: hipcc counts an asm output as written at ;;#ASMEND
and may copy or spill it before the data has landed -- what made bf_f32_regq.hip's first ring wrong once in ~1e5 queries.  This script walks
every kernel's ISA in program order, keeps the queue of LDS operations in flight (asm reads with their destination registers, any other ds_
operation as an anonymous entry; an ASM s_waitcnt lgkmcnt(N) retires all but the youngest N, LDS returns in order -- a compiler wait is
ignored: it may sit in a block the hot path jumps over; asm loads from memory and vmcnt the same way) and
FAILS on a compiler instruction of a copying kind (v_mov*, v_pk_mov*, v_accvgpr_*, v_swap*, scratch_store*) whose SOURCE is a register
still in flight.  (Program order is not execution order across branches: other readers -- an MFMA in the block a prologue jumps over --
show up as artefacts and are not judged; a copy is what the compiler adds on its own, and what went wrong.)
    python tools/check_asm_reads.py [unit ...]        run for bf_bf16_stream by tests/test_build_guards.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COPY = ("v_mov", "v_pk_mov", "v_accvgpr", "v_swap", "scratch_store")


def vregs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def scan(lines, queue, vq):
    """One pass in program order; returns (findings, asm reads seen, (LDS queue, memory queue) at the end)."""
    queue, vq = list(queue), list(vq)
    inasm, nreads, found = False, 0, []
    for l in lines:
        if "#ASMSTART" in l:
            inasm = True
            continue
        if "#ASMEND" in l:
            inasm = False
            continue
        c = l.split(";")[0].strip()
        if not c or c.startswith(".") or c.endswith(":"):
            continue
        if c.startswith("s_waitcnt"):
            if not inasm:
                continue   # a compiler wait may sit in a block the hot path jumps over (the append path's vmcnt(0)): only the asm waits retire
            w = re.search(r"lgkmcnt\((\d+)\)", c)
            if w:
                n = int(w.group(1))
                queue = queue[max(0, len(queue) - n):] if n else []
            w = re.search(r"vmcnt\((\d+)\)", c)
            if w:
                n = int(w.group(1))
                vq = vq[max(0, len(vq) - n):] if n else []
            continue
        vmem = c.startswith(("global_", "buffer_", "flat_", "scratch_"))
        if inasm:
            if c.startswith("ds_read"):
                queue.append(vregs(c.split(",")[0]))
                nreads += 1
            elif c.startswith("ds_"):
                queue.append(set())
            elif vmem and "_load" in c and " lds" not in c and "load_lds" not in c:
                vq.append(vregs(c.split(",")[0]))   # (the same rule for an asm load from memory, counted in vmcnt)
                nreads += 1
            elif vmem:
                vq.append(set())
            continue
        if c.startswith(COPY):
            ops = c.split(None, 1)[1] if " " in c else ""
            srcs = ",".join(ops.split(",")[1:]) if not c.startswith("scratch_store") else ops
            live = set().union(*queue, *vq) if (queue or vq) else set()
            hit = vregs(srcs) & live
            if hit:
                found.append(f"{c}   <- v{min(hit)} is the destination of an asm read still in flight")
        if not c.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "ds_write", "s_", "v_cmp", "v_cmpx")) and " " in c:
            dst = vregs(c.split(None, 1)[1].split(",")[0])   # a compiler WRITE ends the asm value's life in that register: no longer judged
            if dst:
                queue = [q - dst for q in queue]
                vq = [q - dst for q in vq]
        if c.startswith("ds_"):
            queue.append(set())
        elif vmem:
            vq.append(set())
    return found, nreads, (queue, vq)


def check(unit="bf_bf16_stream", verbose=True, src=None):
    src = src or os.path.join(ROOT, "zvdb_amd", "csrc", unit + ".hip")
    asm = f"/tmp/check_asm_reads_{unit}.s"
    extra = ["-ffp-contract=off"] if unit in ("exact_kernels", "ef_search", "rerank", "graph_build", "exact_scan") else []
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "zvdb_amd", "csrc"), *extra, "-S", "--cuda-device-only", src, "-o", asm], check=True, stderr=subprocess.DEVNULL)
    text = open(asm).read().splitlines()
    bad, seen, reads_total = [], 0, 0
    i = 0
    while i < len(text):
        m = re.match(r"^(_Z\S+):\s*(;.*)?$", text[i])
        if not m or not re.search(r"^\s*\.type\s+" + re.escape(m.group(1)) + r",@function", "\n".join(text[max(0, i - 6):i]), re.M):
            i += 1
            continue
        name = m.group(1)
        j = i
        while j < len(text) and "s_endpgm" not in text[j] and not text[j].startswith(".Lfunc_end"):
            j += 1
        body = text[i:j]
        i = j
        # program order first; then every loop body once more with the state its own first pass left behind (a ring's reads are issued at
        # the END of an iteration for the next one: in flight at the loop head only on the second time round)
        found, nreads, _ = scan(body, [], [])
        labels = {l.split(":")[0]: n for n, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        for n, l in enumerate(body):
            mb = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if mb and mb.group(1) in labels and labels[mb.group(1)] < n:
                loop = body[labels[mb.group(1)]:n + 1]
                _, nr, state = scan(loop, [], [])
                if nr:
                    f2, _, _ = scan(loop, *state)
                    found += f2
        for what in dict.fromkeys(found):
            bad.append((name, what))
        if nreads:
            seen += 1
            reads_total += nreads
    if verbose:
        for name, what in bad:
            print("FAIL", name[:70], what)
        print(f"{unit}: {seen} kernels with asm LDS reads ({reads_total} reads), {len(bad)} findings")
    return seen, bad


if __name__ == "__main__":
    units = sys.argv[1:] or ["bf_bf16_stream"]
    total = 0
    for u in units:
        seen, bad = check(u)
        total += len(bad) + (seen == 0)
    sys.exit(1 if total else 0)
