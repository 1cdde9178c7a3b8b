"""Wall time of one insert_leaf / verify_non_inclusion constraint check (host pointers) at depth 32: the quad-per-item
kernels against one thread per chain."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, imt_amd, oracle_lib
for name, coop in (("one thread per chain", 0), ("quad per chain     ", 16384)):
    c = imt_amd.Context(0)
    c.set_option(imt_amd._ffi.OPT_COOP_MAX_EVENTS, coop)
    t = imt_amd.IndexedTree(c, 32, 1 << 12)
    t.insert_batch(oracle_lib.synth_values(100, 5))
    for n in (1, 64):
        r = t.insert_batch(oracle_lib.synth_values(n, 100 + n))
        f = lambda: c.insert_witness(r["old_root"], r["low_leaf"], r["low_index"], r["low_sib"], r["new_root"], r["new_leaf"],
                                     r["new_index"], r["new_sib"], r["is_largest"], 32)
        assert not f().any()
        t0 = time.perf_counter(); f(); dt = time.perf_counter() - t0
        cand = oracle_lib.synth_values(n, 900 + n)
        low, leaves, sib, lg = t.non_membership_witness(cand)
        root = imt_amd.to_bytes(t.root())
        g = lambda: c.non_membership(root, leaves, low, sib, 32, imt_amd.to_bytes(cand), lg)
        assert not g().any()
        t1 = time.perf_counter(); g(); dn = time.perf_counter() - t1
        print(f"{name}  n={n:3d}  insert_leaf check {dt * 1e3:6.2f} ms   verify_non_inclusion check {dn * 1e3:6.2f} ms")
    t.close(); c.close()
