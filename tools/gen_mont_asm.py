#!/usr/bin/env python3
"""Generates indexed-merkle-tree-halo2_amd/csrc/imt_mont_asm.hpp: the Montgomery products of
imt_device.hpp (mont_dot / mont_sqr) as single inline-asm blocks for gfx950.

Why: hipcc re-associates each column sum so that the carry from the previous column is added last
(one extra v_lshl_add_u64 per column, 17 per reduction) to expose instruction-level parallelism.
tools/microbench/valu_rates.hip shows that ONE dependent chain of v_mad_u64_u32 already issues at
the full rate (5.4 cycles per wave instruction at 1 wave/SIMD, 4.2 at 8 -- the same as 8 independent
chains), so the extra adds buy nothing.  The blocks below keep each column a single chain:

    column k (k < 9):   acc += sum a_t[i]*b_t[k-i] ; acc += sum_{i<k} m[i]*p[k-i]
                        m[k] = lo32(acc) * (-p^-1 mod 2^32)      (narrow: & 2^29-1)
                        acc += m[k]*p[0] ; acc >>= 29
    column k (k >= 9):  acc += products ; acc += sum m[i]*p[k-i] ; (acc += addend[k-9])
                        r[k-9] = lo32(acc) & (2^29-1) ; acc >>= 29
    r[8] = lo32(acc) (+ addend[8])

The values are the same as the C++ forms in imt_device.hpp (which the host build and the CPU tests
keep using); tests/test_gpu_parity.py checks the device results against the oracle.

The accumulator lives in a fixed VGPR pair because inline asm has no way to name the low half of
a 64-bit operand.  m[j] and r[j] share a register (m[j] is last read in column j+8, r[j] is written
in column j+9).
"""
import os
import sys

NL = 9
ACC = "v[38:39]"       # caller-saved, even-aligned (64-bit operands must be on gfx950)
ACC_LO = "v38"
MASK = "0x1fffffff"


def gen(name, nt, a_kind, add, wide, sqr=False, doc="", add_kind="v", redc=False):
    """a_kind: 's' = the first factor of every term is a wave-uniform constant held in SGPRs,
    'v' = per-lane values.  The second factor is always per-lane.  add_kind: the same for the addend."""
    lines = []
    ops_out = ["[r%d] \"=&v\"(r%d)" % (j, j) for j in range(NL)]
    ops_in = []
    if sqr or redc:
        for i in range(NL):
            ops_in.append("[a%d] \"v\"(a.v[%d])" % (i, i))
        for i in range(NL - 1 if sqr else 0):
            ops_in.append("[d%d] \"v\"(d%d)" % (i, i))
    else:
        for t in range(nt):
            for i in range(NL):
                ops_in.append("[a%d_%d] \"%s\"(a[%d].v[%d])" % (t, i, a_kind, t, i))
            for i in range(NL):
                ops_in.append("[b%d_%d] \"v\"(b[%d].v[%d])" % (t, i, t, i))
    if add:
        for i in range(NL):
            ops_in.append("[e%d] \"%s\"(addend.v[%d])" % (i, add_kind, i))
    for i in range(NL):
        ops_in.append("[p%d] \"s\"(p29(%d))" % (i, i))
    ops_in.append("[n0] \"s\"(N0INV32)")

    first = [True]

    def mad(x, y):
        src2 = "0" if first[0] else ACC
        first[0] = False
        lines.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC, x, y, src2))

    def products(k):
        lo, hi = max(0, k - (NL - 1)), min(k, NL - 1)
        if redc:                    # the value itself sits in the low nine columns: r = a / R
            if k < NL:
                mad("%%[a%d]" % k, "1")
        elif sqr:
            for i in range(lo, hi + 1):
                if 2 * i < k:
                    mad("%%[d%d]" % i, "%%[a%d]" % (k - i))
            if k % 2 == 0:
                mad("%%[a%d]" % (k // 2), "%%[a%d]" % (k // 2))
        else:
            for t in range(nt):
                for i in range(lo, hi + 1):
                    mad("%%[a%d_%d]" % (t, i), "%%[b%d_%d]" % (t, k - i))

    for k in range(NL):
        products(k)
        for i in range(k):
            mad("%%[r%d]" % i, "%%[p%d]" % (k - i))
        lines.append("v_mul_lo_u32 %%[r%d], %s, %%[n0]" % (k, ACC_LO))
        if not wide:
            lines.append("v_and_b32_e32 %%[r%d], %s, %%[r%d]" % (k, MASK, k))
        mad("%%[r%d]" % k, "%[p0]")
        lines.append("v_lshrrev_b64 %s, 29, %s" % (ACC, ACC))
    for k in range(NL, 2 * NL - 1):
        products(k)
        for i in range(k - (NL - 1), NL):
            mad("%%[r%d]" % i, "%%[p%d]" % (k - i))
        if add:
            mad("%%[e%d]" % (k - NL), "1")
        lines.append("v_and_b32_e32 %%[r%d], %s, %s" % (k - NL, MASK, ACC_LO))
        if k < 2 * NL - 2 or add:
            lines.append("v_lshrrev_b64 %s, 29, %s" % (ACC, ACC))
    if add:
        # the addend first: src0 of a VOP2 may be an SGPR (uniform addend), src1 must be a VGPR
        lines.append("v_add_u32_e32 %%[r%d], %%[e%d], %s" % (NL - 1, NL - 1, ACC_LO))
    else:   # low word of (acc >> 29) straight into the top limb
        lines.append("v_alignbit_b32 %%[r%d], %s, %s, 29" % (NL - 1, "v39", ACC_LO))

    n_mad = sum(1 for l in lines if l.startswith("v_mad_u64_u32"))
    if sqr or redc:
        sig = "Fe& r, const Fe& a"
    else:
        sig = "Fe& r, const Fe* a, const Fe* b" + (", const Fe& addend" if add else "")
    out = []
    out.append("// %s  (%d instructions, %d of them v_mad_u64_u32)" % (doc, len(lines), n_mad))
    out.append("__device__ __forceinline__ void %s(%s) {" % (name, sig))
    out.append("    uint32_t %s;" % ", ".join("r%d" % j for j in range(NL)))
    if sqr:
        out.append("    const uint32_t %s;" % ", ".join("d%d = a.v[%d] << 1" % (i, i) for i in range(NL - 1)))
    out.append("    asm(")
    for l in lines:
        out.append("        \"%s\\n\"" % l)
    out.append("        : " + ", ".join(ops_out))
    # wrap the input list
    out.append("        : " + ",\n          ".join(", ".join(ops_in[i:i + 4]) for i in range(0, len(ops_in), 4)))
    out.append("        : \"vcc\", \"v38\", \"v39\");")
    for j in range(NL):
        out.append("    r.v[%d] = r%d;" % (j, j))
    out.append("}")
    out.append("")
    return "\n".join(out)


HEADER = '''// imt_mont_asm.hpp -- GENERATED by tools/gen_mont_asm.py; do not edit.
// The Montgomery products of imt_device.hpp as single inline-asm blocks for gfx950, one dependent
// v_mad_u64_u32 chain per column (see the generator for why).  Device compilation only; the host
// build (tests/native/emul_device.cpp) and the value semantics are those of the C++ forms.
//
// "_uc" variants take the FIRST factor of every term as wave-uniform constants ("s" constraints:
// they must come from scalar loads of __constant__ data indexed by uniform values -- a per-lane
// value passed there would silently be replaced by lane 0's).
#pragma once
#if !defined(__HIP_DEVICE_COMPILE__)
#error "device-only header"
#endif

namespace imt {
namespace dev {
namespace masm {

'''

FOOTER = '''}  // namespace masm
}  // namespace dev
}  // namespace imt
'''


def main():
    body = [HEADER]
    body.append(gen("mul_vv", 1, "v", False, True, doc="r = a[0]*b[0] / R, wide digits: r < a*b/R + 8p"))
    body.append(gen("sqr_v", 1, "v", False, True, sqr=True, doc="r = a^2 / R, wide digits; limbs of a < 2^30"))
    body.append(gen("dot3_uc", 3, "s", False, True, doc="r = sum_{t<3} a[t]*b[t] / R, a uniform constants, wide digits"))
    body.append(gen("dot4_uc", 4, "s", False, True, doc="r = sum_{t<4} a[t]*b[t] / R, a uniform constants, wide digits"))
    body.append(gen("dot2_add_uc_narrow", 2, "s", True, False,
                    doc="r = (sum_{t<2} a[t]*b[t] + addend*R) / R, a uniform constants, 29-bit digits: r < .../R + addend + p"))
    # the witness-trace kernel (imt_trace_device.hpp): one product per emitted value, 29-bit digits throughout
    body.append(gen("sqr_v_narrow", 1, "v", False, False, sqr=True, doc="r = a^2 / R, 29-bit digits: r < a^2/R + p"))
    body.append(gen("mul_vv_adds_narrow", 1, "v", True, False, add_kind="s",
                    doc="r = (a[0]*b[0] + addend*R) / R, addend a uniform constant, 29-bit digits"))
    body.append(gen("mul_uc_narrow", 1, "s", False, False, doc="r = a[0]*b[0] / R, a a uniform constant, 29-bit digits"))
    body.append(gen("mul_uc_add_narrow", 1, "s", True, False,
                    doc="r = (a[0]*b[0] + addend*R) / R, a a uniform constant, addend per lane, 29-bit digits"))
    body.append(gen("redc_v_narrow", 0, "v", False, False, redc=True,
                    doc="r = a / R (out of the Montgomery domain), 29-bit digits: r < a/R + p"))
    # the lane-cooperative hash for small batches (imt_coop_device.hpp): every factor is a per-lane value, because what
    # is a constant for one lane of a quad is a state value for its neighbour; 29-bit digits throughout
    body.append(gen("mul_vv_narrow", 1, "v", False, False, doc="r = a[0]*b[0] / R, 29-bit digits: r < a*b/R + p"))
    body.append(gen("mul_vv_add_narrow", 1, "v", True, False, doc="r = (a[0]*b[0] + addend*R) / R, all per lane, 29-bit digits"))
    body.append(gen("dot3_vv_narrow", 3, "v", False, False, doc="r = sum_{t<3} a[t]*b[t] / R, all per lane, 29-bit digits"))
    body.append(FOOTER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "indexed-merkle-tree-halo2_amd", "csrc", "imt_mont_asm.hpp")
    if len(sys.argv) > 1:          # tests regenerate into a scratch file and compare
        path = sys.argv[1]
    with open(path, "w") as f:
        f.write("\n".join(body))
    print("wrote", path)


if __name__ == "__main__":
    main()
