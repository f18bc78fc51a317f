"""Static VALU instruction mix of k_extprod<13> (the Params(1024) throughput kernel), from the
gfx950 assembly hipcc produces for the current sources: fractions of 64-bit multiply-adds
(v_mad_u64_u32 / v_mad_i64_i32), 32-bit multiplies (v_mul_lo / v_mul_hi / 24-bit forms), the
full-rate simple instructions (add, sub, two-operand logic, right shifts, moves) and every other
vector ALU instruction.  bench.py prices the kernel's measured VALU instruction count against the
micro-benchmarked issue rates of these four classes (tools/ubench_int.hip).
usage: python tools/valu_mix.py  -> JSON on stdout"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_body(asm, pattern):
    m = re.search(r"^_ZN5sgfhe9%s[^:\n]*:.*?\n(.*?)\n\.Lfunc_end" % pattern, asm, re.S | re.M)
    if not m:
        raise SystemExit("kernel %s not found in the assembly" % pattern)
    return m.group(1)


# Full-rate vector instructions on the MI355X (tools/ubench_int.hip, profiles/r03_ubench_valu.txt:
# 56-63 T lane-operations/s): additions, subtractions, the two-operand logic operations, right
# shifts and moves.  Everything else that is not a multiply (left shifts, min / max, the
# three-operand forms, v_alignbit, bit-field extracts, carry forms) issues at 36-38 T.
FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_ashrrev_i32",
        "v_lshrrev_b32", "v_mov_b32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32")


def mix_of(body):
    ops = [l.split()[0] for l in body.splitlines() if l.strip().startswith("v_")]
    # loop bodies appear once in the text; weights below are static counts, which is what the
    # rolled four-phase loop makes representative (the phase loop is 4 x the same code)
    n = len(ops)
    mad64 = sum(o.startswith(("v_mad_u64_u32", "v_mad_i64_i32")) for o in ops)
    mul = sum(o.startswith(("v_mul_lo", "v_mul_hi", "v_mul_u32_u24", "v_mul_i32_i24", "v_mad_u32_u24",
                            "v_mad_i32_i24")) for o in ops)
    fast = sum(o.split("_e32")[0].split("_e64")[0] in FAST for o in ops)
    return {"mad64": mad64 / n, "mul": mul / n, "fast": fast / n, "slow": (n - mad64 - mul - fast) / n,
            "static_valu": n}


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "engine.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                               "-S", "--cuda-device-only", "-o", out,
                               os.path.join(ROOT, "sgfhe.jl_amd", "csrc", "engine.hip")],
                              stderr=subprocess.DEVNULL)
        asm = open(out).read()
    print(json.dumps(mix_of(kernel_body(asm, "k_extprodILi13ELi4ELb0E"))))


if __name__ == "__main__":
    main()
