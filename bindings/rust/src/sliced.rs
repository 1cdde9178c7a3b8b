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
    /// the live options of this world: `IMT_SLICED_OPT_PREP_STREAM`, `_WATCHDOG_MS`, `_TIMING`.  The others decide which
    /// streams exist and are process-wide defaults for worlds created later: `SlicedTree::set_default_option`
    /// (e.g. `IMT_SLICED_OPT_POOLS` 0 for a host that uses high- or low-priority HIP streams of its own on the device).
    pub fn set_option(&mut self, option: i32, value: i64) -> Result<(), i32> {
        let rc = unsafe { imt_sliced_set_option(self.w, option, value as std::os::raw::c_long) };
        if rc == IMT_OK { Ok(()) } else { Err(rc) }
    }
    pub fn set_default_option(option: i32, value: i64) -> Result<(), i32> {
        let rc = unsafe { imt_sliced_set_option(std::ptr::null_mut(), option, value as std::os::raw::c_long) };
        if rc == IMT_OK { Ok(()) } else { Err(rc) }
    }
    /// where the world stands, as text: what an `IMT_ERR_TIMEOUT` writes to stderr (for a host's own watchdog)
    pub fn dump(&mut self) -> String {
        let n = unsafe { imt_sliced_dump(self.w, std::ptr::null_mut(), 0) };
        let mut buf = vec![0u8; n.max(0) as usize + 1];
        unsafe { imt_sliced_dump(self.w, buf.as_mut_ptr() as *mut std::os::raw::c_char, buf.len()) };
        let end = buf.iter().position(|&b| b == 0).unwrap_or(buf.len());
        String::from_utf8_lossy(&buf[..end]).into_owned()
    }
    /// the subtree layout's one collective on this transport (32 bytes per rank: the subtree roots), when no world uses it
    ///
    /// # Safety
    /// `send` / `recv` are device-addressable, `recv` holds `world * bytes`.
    pub unsafe fn all_gather(tp: *mut imt_transport, send: *const c_void, recv: *mut c_void, bytes: usize, stream: *mut c_void) -> Result<(), i32> {
        let rc = imt_transport_all_gather(tp, send, recv, bytes, stream);
        if rc == IMT_OK { Ok(()) } else { Err(rc) }
    }
    /// after synchronising the stream of an `all_gather` and BEFORE reading `recv`: did a GPU-side wait give up on a peer?
    /// (sticky; `recv` was not written then)
    ///
    /// # Safety
    /// `tp` is a live transport handle.
    pub unsafe fn poll_error(tp: *mut imt_transport) -> Result<(), i32> {
        let rc = imt_transport_poll_error(tp);
        if rc == IMT_OK { Ok(()) } else { Err(rc) }
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
