//! Several GPUs, the reference's single sorted list (`update_idx_leaf`, `src/indexed_merkle_tree.rs:632-660`), time-sliced:
//! what a Rust host needs to drive `imt_itree_slice_*` with its own RCCL (`ncclAllGather`) calls.
//!
//! A step's `world * n` insertions are cut into `world` consecutive slices; GPU g hashes slice g and returns its
//! witnesses; every GPU keeps a replica; each level's write-back travels as a payload.  [`SliceSchedule`] is the
//! arithmetic (the same as `indexed-merkle-tree-halo2_amd/sliced.py` and `imt::SliceSchedule` in `include/imt.hpp`):
//! in round tick `rt` rank `g` runs unit `rt - g * lag` of its slice (0 = leaf hashes, 1 + l = level l -> l + 1), all
//! ranks all-gather that tick's payloads, and the gather of tick `rt` is applied at tick `rt + lag`; consecutive steps
//! (rounds) start `world * lag` ticks apart, at most four in flight, each on its own stream.
//! Ordering the host must keep: round R's unit q, and round R's applies of payloads for unit q, run behind round
//! R - 1's tick `q + world * lag`.
//!
//! All pointers here are DEVICE pointers (or `imt_host_alloc` memory); streams are `hipStream_t` as `*mut c_void`.
//! NOT COMPILED in the repository this file ships in; `tests/test_rust_binding.py` checks the FFI names it uses.

use crate::ffi::*;
use std::os::raw::c_void;

pub const ROUNDS_IN_FLIGHT: usize = 4;

#[derive(Clone, Copy, Debug)]
pub struct SliceSchedule {
    pub world: usize,
    pub units: usize,
    pub lag: usize,
    /// global ticks between the starts of consecutive rounds
    pub period: usize,
    /// round ticks that have a compute phase / a collective
    pub gathers: usize,
    /// + the ticks that only apply
    pub round_ticks: usize,
}

impl SliceSchedule {
    /// `units` = depth + 1; `lag` = None picks the smallest lag that keeps at most four rounds in flight (and >= 2, so
    /// that a gather overlaps the next unit)
    pub fn new(world: usize, units: usize, lag: Option<usize>) -> Option<Self> {
        if world < 1 || units < 2 {
            return None;
        }
        let fit = (units + (ROUNDS_IN_FLIGHT - 1) * world - 1) / ((ROUNDS_IN_FLIGHT - 1) * world);
        let lag = lag.unwrap_or(fit.max(2));
        if lag < 1 {
            return None;
        }
        let period = world * lag;
        let gathers = units + (world - 1) * lag;
        let round_ticks = gathers + lag;
        if (round_ticks + period - 1) / period > ROUNDS_IN_FLIGHT {
            return None;
        }
        Some(SliceSchedule { world, units, lag, period, gathers, round_ticks })
    }
    /// unit rank `rank` computes at round tick `rt`
    pub fn unit_of(&self, rank: usize, rt: usize) -> Option<usize> {
        let q = rt.checked_sub(rank * self.lag)?;
        if q < self.units { Some(q) } else { None }
    }
    /// unit whose payload rank `rank` contributes to the all-gather of round tick `rt` (unit 0 writes nothing back)
    pub fn payload_unit(&self, rank: usize, rt: usize) -> Option<usize> {
        self.unit_of(rank, rt).filter(|q| *q >= 1)
    }
    pub fn has_gather(&self, rt: usize) -> bool {
        rt < self.gathers && (0..self.world).any(|g| self.payload_unit(g, rt).is_some())
    }
}

/// One replica's tree as the slice calls see it.  The caller owns the `imt_itree` (e.g. `gpu::IndexedTree`) and the
/// context's thread.
pub struct SlicedTree {
    pub tree: *mut imt_itree,
    pub depth: usize,
}

impl SlicedTree {
    /// bytes of the largest payload of a slice of n insertions (buffer size)
    pub fn payload_bytes(n: usize) -> usize {
        unsafe { imt_itree_slice_payload_bytes(n) }
    }
    /// bytes the payload of `unit` actually uses: what every rank contributes to that tick's all-gather is the maximum
    /// of this over the ranks that run a unit >= 1 in the tick
    pub fn unit_bytes(&self, size_before: u64, n: usize, unit: usize) -> usize {
        unsafe { imt_itree_slice_unit_bytes(self.tree, size_before, n, unit as u32) }
    }
    /// index work for the whole step (identical values on every GPU), events for the own slice; blocks until the values
    /// are checked.  Returns the slice id.
    ///
    /// # Safety
    /// `vals` must point to `(n_before + n_own + n_after) * 32` bytes of device memory; the pointers in `out` must stay
    /// valid until the slice's last unit has run.
    pub unsafe fn prepare(&self, vals: *const c_void, n_before: usize, n_own: usize, n_after: usize, out: &imt_insert_out, fmt: u32) -> Result<i32, i32> {
        let mut slice = -1;
        let rc = imt_itree_slice_prepare(self.tree, vals, n_before, n_own, n_after, out, IMT_DEVICE_PTRS | fmt, &mut slice, std::ptr::null_mut());
        if rc == IMT_OK { Ok(slice) } else { Err(rc) }
    }
    /// enqueue unit `unit` (0..=depth, in order) of an open slice on `stream`; `payload` receives `payload_bytes(n)` bytes
    ///
    /// # Safety
    /// `payload` is a 16-byte aligned device pointer of that size; `stream` a valid `hipStream_t` or null.
    pub unsafe fn unit(&self, slice: i32, unit: usize, payload: *mut c_void, stream: *mut c_void) -> Result<(), i32> {
        let rc = imt_itree_slice_unit(self.tree, slice, unit as u32, payload, stream);
        if rc == IMT_OK { Ok(()) } else { Err(rc) }
    }
    /// apply the `world` payloads of one all-gather (payload r at `gathered + r * stride`; `units[r] < 0` = skip, e.g. the
    /// own rank)
    ///
    /// # Safety
    /// `gathered` holds `units.len() * stride` bytes of device memory produced by the peers' `unit` calls.
    pub unsafe fn apply_gathered(&self, gathered: *const c_void, stride: usize, size_before: &[u64], n: &[u64], units: &[i32], stream: *mut c_void) -> Result<(), i32> {
        assert!(size_before.len() == units.len() && n.len() == units.len());
        let rc = imt_itree_slice_apply_gathered(self.tree, gathered, stride, units.len(), size_before.as_ptr(), n.as_ptr(), units.as_ptr(), stream);
        if rc == IMT_OK { Ok(()) } else { Err(rc) }
    }
}
