"""PCIe-inclusive insertion rates through the three ways a host program can receive the witnesses
(PCIe note): (1) the Python mirror's default (fresh numpy arrays per call, plain host
pointers, synchronous), (2) plain pageable host pointers with buffers allocated and touched once,
(3) imt_host_alloc buffers passed as device pointers, pipelined."""
import ctypes, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import imt_amd, oracle_lib
from imt_amd import _ffi

lib = imt_amd.lib
ctx = imt_amd.Context(0)
n, depth, nb = 1 << 16, 32, 6
vals = imt_amd.to_bytes(oracle_lib.synth_values(nb * n, 0x494D5402))

t = imt_amd.IndexedTree(ctx, depth, 1 << 20)
t.insert_batch(vals[:n])
ts = []
for i in range(1, nb):
    t0 = time.perf_counter(); t.insert_batch(vals[i * n:(i + 1) * n]); ts.append(time.perf_counter() - t0)
print("(1) python mirror, fresh numpy arrays per call : ms/batch", [round(x * 1e3, 1) for x in ts],
      "-> %.2f M insertions/s" % (n / min(ts) / 1e6))
t.close()


def bufs(alloc):
    return dict(low_index=alloc(n, np.uint64), is_largest=alloc(n, np.uint8), low_leaf=alloc((n, 3, 32), np.uint8),
                new_leaf=alloc((n, 3, 32), np.uint8), old_root=alloc((n, 32), np.uint8),
                interim_root=alloc((n, 32), np.uint8), new_root=alloc((n, 32), np.uint8),
                low_sib=alloc((depth, n, 32), np.uint8), new_sib=alloc((depth, n, 32), np.uint8))


t = imt_amd.IndexedTree(ctx, depth, 1 << 20)
o = bufs(lambda s, d: np.zeros(s, d))
st = _ffi.InsertOut(**{k: v.ctypes.data for k, v in o.items()})
ts = []
for i in range(nb):
    t0 = time.perf_counter()
    rc = lib.imt_itree_insert_batch(t.h, ctypes.c_void_p(vals.ctypes.data + i * n * 32), n, ctypes.byref(st), 0)
    assert rc == 0
    ts.append(time.perf_counter() - t0)
print("(2) pageable host pointers, buffers reused     : ms/batch", [round(x * 1e3, 1) for x in ts[1:]],
      "-> %.2f M insertions/s" % (n / min(ts[1:]) / 1e6))
root2 = t.root()
t.close()

t = imt_amd.IndexedTree(ctx, depth, 1 << 20)
pv = ctx.host_alloc((nb * n, 32)); pv[:] = vals
sets = [bufs(lambda s, d: ctx.host_alloc(s, d)) for _ in range(3)]     # as many as batches in flight
t0 = time.perf_counter()
for i in range(nb):
    if i >= 3:
        pass      # a real caller consumes set i % 3 here, after the sync that covers batch i-3
    st = _ffi.InsertOut(**{k: v.ctypes.data for k, v in sets[i % 3].items()})
    rc = lib.imt_itree_insert_batch(t.h, ctypes.c_void_p(pv.ctypes.data + i * n * 32), n, ctypes.byref(st),
                                    _ffi.DEVICE_PTRS | _ffi.PIPELINE)
    assert rc == 0
ctx.sync()
dt = time.perf_counter() - t0
print("(3) imt_host_alloc buffers as device pointers  : %.1f ms/batch over %d pipelined batches -> %.2f M insertions/s"
      % (dt / nb * 1e3, nb, nb * n / dt / 1e6))
assert t.root() == root2
