"""CPU: the product's host logic and a host build of its device arithmetic, against the oracle.
No compute call reaches the GPU library here (there is no GPU): we only check that it loads,
exports every symbol include/imt.h declares, and refuses to run without a device."""
import ctypes
import os
import random
import re

import numpy as np
import pytest

import oracle_lib
from oracle_lib import P, b32

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(imt):
    hdr = open(os.path.join(ROOT, "include", "imt.h")).read()
    declared = set(re.findall(r"\b(imt_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(imt.lib, name), f"libimt_hip.so does not export {name}"
    assert declared == set(imt._ffi.SIGNATURES), declared ^ set(imt._ffi.SIGNATURES)
    assert imt.lib.imt_version().startswith(b"imt-hip gfx950")


def test_no_cpu_fallback(imt):
    import torch
    h = ctypes.c_void_p()
    assert imt.lib.imt_ctx_create(-1, ctypes.byref(h)) == imt._ffi.ERR["NO_DEVICE"]
    if not torch.cuda.is_available():
        with pytest.raises(imt.ImtError) as ei:
            imt.Context(0)
        assert ei.value.code == imt._ffi.ERR["NO_DEVICE"]


def _emul_hash(emul, xs, fi=0, fo=0):
    out = ctypes.create_string_buffer(32)
    rc = emul.emul_hash(b"".join(b32(x) for x in xs), len(xs), out, fi, fo)
    return rc, int.from_bytes(out.raw, "little")


def test_device_field_arithmetic_on_host(emul):
    rng = random.Random(1)
    cases = [(0, 0), (P - 1, P - 1), (1, P - 1), (P - 1, 2), ((1 << 253) + 12345, P - 7)]
    cases += [(rng.randrange(P), rng.randrange(P)) for _ in range(3000)]
    out = ctypes.create_string_buffer(32)
    for x, y in cases:
        emul.emul_mul(b32(x), b32(y), out, 0)
        assert int.from_bytes(out.raw, "little") == x * y % P
        emul.emul_mul(b32(x), b32(y), out, 1)
        assert int.from_bytes(out.raw, "little") == x * x % P


def test_column_accumulator_bounds():
    """Worst-case proof of the no-carry column accumulation in csrc/imt_device.hpp (mont_dot /
    mont_sqr): with 32-bit quotient digits every 64-bit column stays below 2^64, and with the value
    growth of +8p per wide reduction (+p per narrow one) every lane of the permutation stays below
    2^261 with a top limb below 2^29."""
    NL, W = 9, 29
    pl = [(P >> (W * i)) & ((1 << W) - 1) for i in range(NL)]
    assert sum(pl[i] << (W * i) for i in range(NL)) == P
    carry_in = 1 << 36

    def worst_column(ab_terms, m_max):
        # ab_terms(k) = bound of the a*b part of column k; the m*p part holds each limb of p at most once
        worst = 0
        for k in range(2 * NL - 1):
            lo, hi = max(0, k - (NL - 1)), min(k, NL - 1)
            mp = sum(m_max * pl[k - i] for i in range(lo, hi + 1))
            worst = max(worst, ab_terms(k) + mp + carry_in)
        return worst

    def n_products(k):
        return min(k, NL - 1) - max(0, k - (NL - 1)) + 1

    wide, narrow = (1 << 32) - 1, (1 << 29) - 1
    l29, l30, l31 = (1 << 29) - 1, (1 << 30) - 1, (1 << 31) - 1
    # mont_dot<NT>: operands with 29-bit limbs (constants x state), wide digits, NT up to 4
    for nt in (1, 2, 3, 4):
        assert worst_column(lambda k: nt * n_products(k) * l29 * l29, wide) < 1 << 64
    # mont_mul(x4, x): one side lazily added (30-bit limbs)
    assert worst_column(lambda k: n_products(k) * l29 * l30, wide) < 1 << 64
    # mont_sqr of a lazily added value: doubled operand 31 bits x 30 bits, plus the square term
    def sqr_terms(k):
        lo = max(0, k - (NL - 1))
        cross = len([i for i in range(lo, NL) if 2 * i < k])
        return cross * l31 * l30 + (l30 * l30 if k % 2 == 0 else 0)
    assert worst_column(sqr_terms, wide) < 1 << 64
    # the linear-lane update REDC(s R + c y + c' y') keeps 29-bit digits; the addend limb is < 2^32
    assert worst_column(lambda k: 2 * n_products(k) * l29 * l29 + (1 << 32), narrow) < 1 << 64

    # value growth through one permutation, in units of p (R / p > 169)
    R_over_p = (1 << 261) / P
    def red(t_over_p2, wide_m=True):          # bound of REDC(T) given T <= t_over_p2 * p^2
        return t_over_p2 / R_over_p + (8 if wide_m else 1)
    def sbox(x):                              # x in units of p, after the lazy add of a constant
        x2 = red(x * x); x4 = red(x2 * x2); return red(x4 * x)
    lanes = [16.0, 16.0, 16.0]                # entry bound
    worst_lane = max(lanes)
    for _ in range(4):
        y = [sbox(v + 1) for v in lanes]
        lanes = [red(sum(y))] * 3             # matrix entries < p
        worst_lane = max(worst_lane, *lanes)
    s0, s1, s2 = lanes
    for _ in range(29):
        y0 = sbox(s0 + 1)
        n0 = red(y0 + s1 + s2)
        y1 = sbox(n0 + 1)
        n0 = red(y1 + s1 + s2 + y0)
        s1 = s1 + (y0 + y1) / R_over_p + 1
        s2 = s2 + (y0 + y1) / R_over_p + 1
        s0 = n0
        worst_lane = max(worst_lane, s0, s1, s2, y0 + 1, y1 + 1)
    lanes = [s0, s1, s2]
    for _ in range(4):
        y = [sbox(v + 1) for v in lanes]
        lanes = [red(sum(y))] * 3
        worst_lane = max(worst_lane, *lanes)
    assert worst_lane < 64                    # << 169: top limb < 64p >> 232 < 2^28
    assert (64 * P) >> 232 < 1 << 28
    assert max(lanes) < 9                     # what canonicalize (< 16p) and the next block's entry need


def test_generated_assembly_header_is_current(tmp_path):
    """csrc/imt_mont_asm.hpp is generated; the committed copy must be what the generator writes, and
    its instruction counts must be the single-chain minimum (no per-column 64-bit adds)."""
    import subprocess, sys
    out = tmp_path / "imt_mont_asm.hpp"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_mont_asm.py"), str(out)], check=True,
                   stdout=subprocess.DEVNULL)
    committed = open(os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "csrc", "imt_mont_asm.hpp")).read()
    assert out.read_text() == committed
    blocks = committed.split("__device__ __forceinline__ void ")[1:]
    want = {"mul_vv": 81 + 81, "sqr_v": 45 + 81, "dot3_uc": 243 + 81, "dot4_uc": 324 + 81,
            "dot2_add_uc_narrow": 162 + 81 + 8,
            # the witness-trace kernel's one-product forms; "+ 8": the addend limbs enter as mad(e, 1)
            "sqr_v_narrow": 45 + 81, "mul_vv_adds_narrow": 81 + 81 + 8, "mul_uc_narrow": 81 + 81,
            "mul_uc_add_narrow": 81 + 81 + 8,
            "redc_v_narrow": 9 + 81,       # a / R: the nine limbs enter as mad(a, 1), no limb products
            # the lane-cooperative hash
            "mul_vv_narrow": 81 + 81, "mul_vv_add_narrow": 81 + 81 + 8, "dot3_vv_narrow": 243 + 81}
    for b in blocks:
        name = b.split("(")[0]
        assert b.count('"v_mad_u64_u32') == want[name]
        assert "v_lshl_add_u64" not in b and "s_nop" not in b


def test_device_poseidon_on_host_matches_oracle(emul, oracle):
    rng = random.Random(2)
    cases = [[0, 0], [0, 0, 0], [1, 2], [1, 2, 3], [P - 1, P - 1], [P - 1, P - 1, P - 1]]
    cases += [[rng.randrange(P) for _ in range(rng.choice([2, 3]))] for _ in range(200)]
    for c in cases:
        rc, h = _emul_hash(emul, c)
        assert rc == 0 and h == oracle.hash(c)
        out = ctypes.create_string_buffer(32)
        emul.emul_host_hash(b"".join(b32(x) for x in c), len(c), out)        # the table generator's own hash
        assert int.from_bytes(out.raw, "little") == h
    for _ in range(20):
        st = [rng.randrange(P) for _ in range(3)]
        o1 = ctypes.create_string_buffer(96)
        emul.emul_permute(b"".join(map(b32, st)), o1)
        assert [int.from_bytes(o1.raw[32 * i:32 * i + 32], "little") for i in range(3)] == oracle.permute(st)


def test_boundary_formats_on_host(emul, oracle):
    rng = random.Random(3)
    R = 1 << 256
    for _ in range(40):
        c = [rng.randrange(P) for _ in range(2)]
        rc, h = _emul_hash(emul, [x * R % P for x in c], 1, 1)       # halo2curves Montgomery in and out
        assert rc == 0 and h == oracle.hash(c) * R % P
        out = ctypes.create_string_buffer(32)
        emul.emul_convert(b32(c[0]), out, 0, 1)
        assert int.from_bytes(out.raw, "little") == c[0] * R % P
        emul.emul_convert(b32(c[0]), out, 0, 2)                       # device format: R = 2^261, reduced
        assert int.from_bytes(out.raw, "little") == (c[0] << 261) % P
    assert _emul_hash(emul, [P, 0])[0] == -5                          # non-canonical input is flagged
    assert _emul_hash(emul, [(1 << 256) - 1, 0])[0] == -5


def _sweep_reference(events, base_fn, depth_levels):
    """Brute force: replay events in time order on a dict tree; returns per level the list of
    (node, time, sibling_source_time or None) for every event."""
    latest = [dict() for _ in range(depth_levels + 1)]   # node -> time of latest version
    per_level = [[] for _ in range(depth_levels)]
    for t, pos in events:
        latest[0][pos] = t
        for l in range(depth_levels):
            n = pos >> l
            per_level[l].append((n, t, latest[l].get(n ^ 1)))
            latest[l + 1][n >> 1] = t
    return per_level


def test_merge_element_against_bruteforce(emul):
    rng = random.Random(4)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    for trial in range(30):
        n_ins = rng.randrange(1, 40)
        M = rng.randrange(1, 20)
        events = []
        for i in range(n_ins):
            events.append((2 * i, rng.randrange(0, M + i)))     # low leaf: any earlier leaf
            events.append((2 * i + 1, M + i))
        total = len(events)
        L0 = max(1, (M + n_ins - 1).bit_length())
        ref = _sweep_reference(events, None, L0)
        keys = sorted((pos, t) for t, pos in events)
        node = np.array([k[0] for k in keys], np.uint32)
        time = np.array([k[1] for k in keys], np.uint32)
        rs = np.zeros(total, np.uint32); re_ = np.zeros(total, np.uint32)
        k = 0
        while k < total:
            j = k
            while j < total and node[j] == node[k]:
                j += 1
            rs[k:j] = k; re_[k:j] = j
            k = j
        for l in range(L0):
            o = [np.zeros(total, np.uint32) for _ in range(5)]
            sib = np.zeros(total, np.int32); nb = np.zeros(total, np.uint32)
            emul.emul_merge_level(node.ctypes.data_as(u32p), time.ctypes.data_as(u32p), rs.ctypes.data_as(u32p),
                                  re_.ctypes.data_as(u32p), total, *[a.ctypes.data_as(u32p) for a in o],
                                  sib.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), nb.ctypes.data_as(u32p))
            n2, t2, rs2, re2, frm = o
            # next table is sorted by (node, time) with correct runs
            order = sorted(range(total), key=lambda x: (int(node[x]) >> 1, int(time[x])))
            assert [int(x) for x in frm & 0x7fffffff] == order
            assert all(int(n2[i]) == int(node[order[i]]) >> 1 and int(t2[i]) == int(time[order[i]]) for i in range(total))
            for i in range(total):
                assert n2[rs2[i]] == n2[i] and n2[re2[i] - 1] == n2[i]
                assert rs2[i] == 0 or n2[rs2[i] - 1] != n2[i]
                assert re2[i] == total or n2[re2[i]] != n2[i]
            # sibling source = newest older version of the sibling node, per the brute force
            want = {(n, t): s for n, t, s in ref[l]}
            for i in range(total):
                src = int(sib[i])
                got = None if src < 0 else int(time[src])
                assert int(nb[i]) == int(node[order[i]])
                assert got == want[(int(nb[i]), int(t2[i]))]
                if src >= 0:
                    assert int(node[src]) == int(nb[i]) ^ 1
                last = bool(frm[i] >> 31)
                kk = int(frm[i] & 0x7fffffff)
                assert last == (kk == int(re_[kk]) - 1)
            node, time, rs, re_ = n2, t2, rs2, re2
        assert (node == 0).all() and [int(x) for x in time] == list(range(total))


def test_synth_values_are_valid():
    v = oracle_lib.synth_values(500, 0x494D5402)
    assert len(set(v)) == 500 and all(0 < x < P for x in v)
    assert v == oracle_lib.synth_values(500, 0x494D5402)


def test_prepare_logic_nearest_smaller_and_count_below(emul):
    """imt_prep_logic.hpp: the sparse-table nearest-smaller searches and the 256-bit lower bound."""
    rng = random.Random(8)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    for n in (1, 2, 3, 7, 8, 9, 100, 257):
        for trial in range(4):
            t = list(range(n))
            if trial == 1:
                t.reverse()
            elif trial >= 2:
                rng.shuffle(t)
            levels = 1
            while (1 << levels) <= n:
                levels += 1
            st = np.zeros(levels * n, np.uint32)
            st[:n] = t
            emul.emul_sparse_table(st.ctypes.data_as(u32p), n, levels)
            for j in range(n):
                want_l = next((i for i in range(j - 1, -1, -1) if t[i] < t[j]), n)
                want_r = next((i for i in range(j + 1, n) if t[i] < t[j]), n)
                assert emul.emul_nsl(st.ctypes.data_as(u32p), n, levels, j) == want_l
                assert emul.emul_nsr(st.ctypes.data_as(u32p), n, levels, j) == want_r
    vals = sorted(rng.sample(range(1, 10 ** 6), 50)) + [(5 << 200) + 3, (5 << 200) + 9, P - 1]
    vals = [0] + sorted(vals)
    perm = list(range(len(vals)))
    rng.shuffle(perm)                                  # leaf order != value order
    store = [0] * len(vals)
    for rank, leaf in enumerate(perm):
        store[leaf] = vals[rank]
    val = np.frombuffer(b"".join(b32(x) for x in store), np.uint8).copy()
    srt = np.array(perm, np.uint32)
    for x in vals + [1, 77, (5 << 200) + 4, P - 2, 10 ** 6 + 1]:
        got = emul.emul_count_below(val.ctypes.data_as(ctypes.c_void_p), srt.ctypes.data_as(u32p), len(vals), b32(x))
        assert got == sum(1 for y in vals if y < x)


def test_snapshot_check_logic(emul, oracle):
    """imt_itree_load's per-rank list check (imt_prep_logic.hpp, run by k_load_check) against the oracle's own leaves
    (update_idx_leaf, /root/reference/src/indexed_merkle_tree.rs:632-660) and against planted corruptions."""
    u32p = ctypes.POINTER(ctypes.c_uint32)
    n, base = 200, 5 << 32
    vals = oracle_lib.synth_values(n - 1, 0x4C4F41)
    # two values that agree in their top 64 bits: the radix order leaves them in leaf order
    top = (vals[10] >> 192) << 192
    vals[20] = top | 5
    vals[30] = top | 3
    leaves = [[0, 0, 0]]
    for v in vals:                                   # the reference's sequential insertion
        lo = max((l for l in leaves if l[0] < v), key=lambda l: l[0])
        leaves.append([v, lo[1], lo[2]])
        lo[1], lo[2] = v, base + len(leaves) - 1
    pre = np.frombuffer(b"".join(b32(x) for l in leaves for x in l), dtype=np.uint8).reshape(n, 3, 32).copy()
    order = np.array(sorted(range(n), key=lambda i: leaves[i][0]), dtype=np.uint32)
    coarse = np.array(sorted(range(n), key=lambda i: (leaves[i][0] >> 192, i)), dtype=np.uint32)
    run = lambda p, idx: emul.emul_load_check(p.ctypes.data_as(ctypes.c_void_p), n, ctypes.c_uint64(base), idx.ctypes.data_as(u32p))
    SENT, LINK, LAST, TIE, DUP = 16, 32, 64, 128, 4
    assert run(pre, order) == 0
    assert (coarse != order).any() and run(pre, coarse) & TIE          # leaf 21 (..|5) before leaf 31 (..|3)
    bad = pre.copy(); bad[3, 1, 0] ^= 1;   assert run(bad, order) == LINK      # next_val
    bad = pre.copy(); bad[3, 2, 0] ^= 1;   assert run(bad, order) == LINK      # next_idx
    bad = pre.copy(); bad[7, 2, 20] = 1;   assert run(bad, order) == LINK      # next_idx beyond 64 bits
    bad = pre.copy(); bad[0, 0, 0] = 1;    assert run(bad, order) & SENT
    last = int(order[-1])
    bad = pre.copy(); bad[last, 2, 0] = 1; assert run(bad, order) == LAST
    bad = pre.copy(); bad[last, 1, 0] = 1; assert run(bad, order) == LAST
    a, b = int(order[50]), int(order[51])
    bad = pre.copy(); bad[b, 0] = bad[a, 0]; assert run(bad, order) & DUP
    # without the index base the pointers of a placed subtree do not check
    assert emul.emul_load_check(pre.ctypes.data_as(ctypes.c_void_p), n, ctypes.c_uint64(0), order.ctypes.data_as(u32p)) & LINK


def test_header_is_plain_c_and_example_links():
    """include/imt.h must compile as C11 (it is the FFI contract) and the C example must link against
    the library (no compute: there is no GPU here)."""
    import subprocess
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                        os.path.join(ROOT, "include", "imt.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for name in ("insert_demo", "slice_demo", "sliced_procs_demo"):
        exe = os.path.join(ROOT, "examples", name)
        r = subprocess.run(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Wextra", "-pedantic", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "examples", name + ".c"), "-L",
                            os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "csrc"), "-limt_hip",
                            "-Wl,-rpath," + os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "csrc"), "-o", exe],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    # the multi-process host fails loudly without a GPU (every rank stops at context creation: no CPU path)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([os.path.join(ROOT, "examples", "sliced_procs_demo"), "2", "ipc", "2", "8"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 3 and "imt_ctx_create: -7" in r.stderr and "FAILED" in r.stdout


def test_cpp_host_side_builds_and_fails_loudly_without_a_gpu(tmp_path):
    """include/imt.hpp (the compiled-language mirror of the reference's host types) and the re-enacted reference
    tests build with g++ alone, warnings as errors.  Run without a GPU the program must stop at context creation with
    IMT_ERR_NO_DEVICE: no hash is ever computed on the CPU.  (With a GPU it simply passes.)"""
    import subprocess
    import torch
    csrc = os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "csrc")
    exe = str(tmp_path / "reference_tests")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "native", "reference_tests.cpp"), "-L", csrc, "-limt_hip",
                        "-Wl,-rpath," + csrc, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if torch.cuda.is_available():
        return
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "imt::Error -7" in r.stderr and "no CPU path" in r.stderr
    assert "hash_zero" not in r.stdout


def test_cpp_sliced_wrapper_compiles_against_the_header(tmp_path):
    """include/imt.hpp's imt::Sliced (RAII over imt_sliced_create / _step / _wait / _flush, the multi-GPU single list
    behind one call per step) builds warning-free with g++ against imt.h and links against libimt_hip.so; without a GPU it
    fails where the library says it has no CPU path.  (The schedule itself is tested in tests/test_sliced_schedule.py.)"""
    import subprocess
    src = tmp_path / "sl.cpp"
    src.write_text('''#include "imt.hpp"
#include <cstdio>
int main() {
    try {
        imt::Context c0(0), c1(0);
        imt::IndexedTree a(c0, 32, 256), b(c1, 32, 256);
        imt::Sliced w = imt::Sliced::local({a.get(), b.get()}, 8);
        std::printf("lag %d period %d\\n", w.info().lag, w.info().period);
        w.flush();
    } catch (const imt::Error& e) { std::printf("imt::Error %d\\n", e.code()); return 2; }
    return 0;
}
''')
    exe = str(tmp_path / "sl")
    csrc = os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        str(src), "-L", csrc, "-limt_hip", "-Wl,-rpath," + csrc, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0 and "lag 6 period 12" in r.stdout, r.stdout + r.stderr
    else:
        assert r.returncode == 2 and "imt::Error -7" in r.stdout


def test_bench_synthetic_values_are_field_elements_of_the_right_residue():
    """bench.synth_values (vectorised since round 3): 0 < v < p, v mod N == rank for the subtree leg, no repeats; the
    helper the harness and the size-of-bench GPU tests share"""
    import bench
    for modulus, residue in ((1, 0), (2, 1), (8, 5)):
        a = bench.synth_values(20000, residue, modulus, 7 + modulus)
        assert a.shape == (20000, 32) and a.dtype == np.uint8
        vals = [int.from_bytes(r.tobytes(), "little") for r in a]
        assert all(0 < v < bench.P for v in vals) and all(v % modulus == residue for v in vals)
        assert len(set(vals)) == len(vals)
        assert max(vals) > bench.P // 2 and min(vals) < bench.P // 8          # spread over the field, not clustered
    assert (bench.synth_values(100, 0, 1, 3) == bench.synth_values(100, 0, 1, 3)).all()       # seeded: every rank draws the same step


def test_owner_subtree_constraint_needs_the_canonical_decomposition():
    """bindings/rust/src/chip.rs::constrain_owner_subtree (the one extra constraint of the subtree layout), as integers:
    the field equation q 2^128 + r = v alone is satisfied by the limbs of v AND of v + p for every v < 2^254 - p, and p
    being odd the two disagree about v mod 2^k; the lexicographic bound (q, r) <= limbs(p - 1) keeps exactly the honest one.
    The Rust source carries both conditions and the negative MockProver test."""
    import random
    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    M = (1 << 128) - 1
    pq, pr = (P - 1) >> 128, (P - 1) & M

    def accepted(v, q, r, canonical):
        ok = q < (1 << 128) and r < (1 << 128) and (q * (1 << 128) + r) % P == v % P
        if canonical:
            ok = ok and (q < pq or (q == pq and r < pr + 1))
        return ok
    rng = random.Random(9)
    for v in [10, 1, 0, (1 << 254) - P - 1] + [rng.randrange((1 << 254) - P) for _ in range(200)]:
        honest, forged = (v >> 128, v & M), ((v + P) >> 128, (v + P) & M)
        assert accepted(v, *honest, False) and accepted(v, *forged, False)          # the hole
        assert (v & 7) != ((v + P) & 7)                                              # ... names another subtree
        assert accepted(v, *honest, True) and not accepted(v, *forged, True)         # closed
    for v in [P - 1, P - 2, rng.randrange(P)]:
        assert accepted(v, v >> 128, v & M, True)
    src = open(os.path.join(ROOT, "bindings", "rust", "src", "chip.rs")).read()
    assert "range.range_check(ctx, q, 128)" in src and "gate.assert_is_const(ctx, &canonical, &F::ONE)" in src
    assert "range.div_mod(ctx, r, &one << k, 128)" in src and "div_mod(ctx, *value" not in src
    t = open(os.path.join(ROOT, "bindings", "rust", "tests", "mockprover.rs")).read()
    assert "owner_constraint_refuses_the_limbs_of_v_plus_p" in t and "expect_satisfied(false)" in t
