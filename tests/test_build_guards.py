"""Build-time guards that need no GPU (hipcc cross-compiles gfx950 here)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_f32_regq_group_loop_keeps_its_shape_in_every_instantiation():
    """tools/check_ring_regs.py on bf_f32_regq.hip: inside the group loop of every k_f32_regq instantiation the ring waits are counted
    (no s_waitcnt vmcnt(0)), no ring register is copied, nothing is spilled, 4 MFMAs per ring load.  (The loads are compiler-visible, so
    this guards SPEED; the copies it looks for are what made the round's first, inline-asm form of the ring wrong once in ~10^5 queries.)"""
    spec = importlib.util.spec_from_file_location("check_ring_regs", os.path.join(ROOT, "tools", "check_ring_regs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seen, bad = mod.check(verbose=False)
    assert seen == 28 and not bad, bad[:5]


import pytest


@pytest.mark.parametrize("unit", ["bf_bf16_stream", "bf_i8_stream", "bf_fp4_stream", "bruteforce", "bf_f32_regq"])
def test_no_compiler_copy_of_an_asm_read_in_flight(unit):
    """tools/check_asm_reads.py: the screens' operand rings are inline-asm reads on compiler-allocated registers with separate tied waits;
    hipcc may copy such a register before the data has landed (bf_f32_regq.hip's first ring: one lost neighbour in ~10^5 queries, and this
    audit flags 16 copies in 8 kernels of that revision).  Every kernel of the unit: no copying instruction reads a register whose asm read is
    still in flight.  (The headline scan is bf_bf16_stream; the empirical side is tools/soak_routes.py on the GPU.)"""
    spec = importlib.util.spec_from_file_location("check_asm_reads", os.path.join(ROOT, "tools", "check_asm_reads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seen, bad = mod.check(unit, verbose=False)
    assert seen >= 14 and not bad, bad[:5]
