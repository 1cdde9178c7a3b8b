//! Several GPUs, the reference's single sorted list (`update_idx_leaf`, `src/indexed_merkle_tree.rs:632-660`; insertion i at
//! leaf `size + i`, `:715`), bit-exact with one GPU: a step's `world * n` insertions are cut into `world` consecutive
//! slices, GPU g hashes slice g and returns its witnesses, every GPU keeps a replica.
//!
//! The schedule, its streams and events and the RCCL all-gather live in libimt_hip.so (`imt_sliced_*`, include/imt.h): a
//! host makes ONE call per step.  This file is the safe-ish face of those calls.
//! NOT COMPILED in the repository this file ships in; `tests/test_rust_binding.py` checks the FFI names it uses.

use crate::ffi::*;
use std::os::raw::c_void;

/// One rank (one process per GPU) of a sliced tree over RCCL.
pub struct SlicedTree {
    w: *mut imt_sliced,
    tp: *mut imt_transport,
}

impl SlicedTree {
    /// rank 0 makes `n_comms` ids (1..=IMT_SLICED_ROUNDS) and the host broadcasts them by whatever means it has (MPI, a
    /// socket, a file): the usual NCCL bootstrap
    pub fn unique_ids(n_comms: usize) -> Result<Vec<u8>, i32> {
        let mut ids = vec![0u8; n_comms * IMT_RCCL_UNIQUE_ID_BYTES];
        for i in 0..n_comms {
            let rc = unsafe { imt_rccl_get_unique_id(ids[i * IMT_RCCL_UNIQUE_ID_BYTES..].as_mut_ptr() as *mut c_void) };
            if rc != IMT_OK {
                return Err(rc);
            }
        }
        Ok(ids)
    }

    /// collective: returns when every rank has joined.
    ///
    /// # Safety
    /// `ctx` / `tree` are live handles of this process's GPU; `tree` is empty or equal on every rank.
    pub unsafe fn new(ctx: *mut imt_ctx, tree: *mut imt_itree, ids: &[u8], world: usize, rank: usize, max_slice: usize) -> Result<Self, i32> {
        let mut tp = std::ptr::null_mut();
        let n_comms = (ids.len() / IMT_RCCL_UNIQUE_ID_BYTES) as i32;
        let rc = imt_transport_rccl_create(ctx, ids.as_ptr() as *const c_void, n_comms, world as i32, rank as i32, &mut tp);
        if rc != IMT_OK {
            return Err(rc);
        }
        let mut w = std::ptr::null_mut();
        let trees = [tree];
        let rc = imt_sliced_create(trees.as_ptr(), 1, world as i32, rank as i32, tp, max_slice, 0, &mut w);
        if rc != IMT_OK {
            imt_transport_destroy(tp);
            return Err(rc);
        }
        Ok(SlicedTree { w, tp })
    }

    /// one step: `vals` = all `world * n` values (device memory, identical on every rank); `out` = where THIS rank's
    /// witnesses (insertions `[rank * n, (rank + 1) * n)` of the step) go.  Returns the round number.
    ///
    /// # Safety
    /// `vals` and every pointer in `out` stay valid and untouched until `wait(round)`.
    pub unsafe fn step(&mut self, vals: *const c_void, n: usize, out: &imt_insert_out, flags: u32) -> Result<u64, i32> {
        let mut round = 0u64;
        let rc = imt_sliced_step(self.w, vals, n, out, flags, &mut round);
        if rc == IMT_OK { Ok(round) } else { Err(rc) }
    }
    pub fn wait(&mut self, round: u64) -> Result<(), i32> {
        let rc = unsafe { imt_sliced_wait(self.w, 0, round) };
        if rc == IMT_OK { Ok(()) } else { Err(rc) }
    }
    pub fn flush(&mut self) -> Result<(), i32> {
        let rc = unsafe { imt_sliced_flush(self.w) };
        if rc == IMT_OK { Ok(()) } else { Err(rc) }
    }
    pub fn info(&self) -> imt_sliced_info {
        let mut o = imt_sliced_info::default();
        unsafe { imt_sliced_get_info(self.w, &mut o) };
        o
    }
}

impl Drop for SlicedTree {
    fn drop(&mut self) {
        unsafe {
            imt_sliced_destroy(self.w);
            imt_transport_destroy(self.tp);
        }
    }
}
