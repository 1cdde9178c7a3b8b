// imt_itree_internal.hpp -- what other translation units of the library may know about an imt_itree (imt_itree.cpp).
#pragma once
#include "imt_ctx.hpp"

imt_ctx* imt_itree_ctx(const imt_itree* t);
unsigned imt_itree_depth(const imt_itree* t);
bool imt_itree_is_plain(const imt_itree* t);     // not placed, not partitioned, no sharded batch open
// milliseconds the host has spent waiting for the GPU inside imt_itree_slice_prepare (the values check, plan-set
// back-pressure) since the last call; reset by the call
double imt_itree_take_wait_ms(imt_itree* t);
// an imt_sliced world marks its replicas while steps are in flight; the ordinary imt_itree_* entry points then refuse
void imt_itree_mark_sliced(imt_itree* t, bool busy);
// the stream the NEXT imt_itree_slice_prepare calls enqueue their work on (NULL: the tree's side stream, the default)
void imt_itree_set_slice_prep_stream(imt_itree* t, void* hip_stream);
// the sticky error word of the world's transport (device-visible, may be NULL): applies of gathered payloads are skipped
// once it is non-zero (a GPU-side wait for a peer gave up: the payloads are not there)
void imt_itree_set_slice_poison(imt_itree* t, const uint32_t* device_word);
// > 0: the host waits inside imt_itree_slice_prepare return IMT_ERR_TIMEOUT after this many milliseconds
void imt_itree_set_slice_wait_limit(imt_itree* t, double ms);
// One shot: the NEXT imt_itree_slice_unit / imt_itree_slice_apply_gathered makes its last kernel signal `hip_event` when that
// kernel is the last thing the call enqueues (the pack of a level below l0; the apply of at least one payload) --
// imt_itree_take_slice_tail_attached then says true, and the caller needs no hipEventRecord behind the call.
void imt_itree_set_slice_tail_event(imt_itree* t, void* hip_event);
bool imt_itree_take_slice_tail_attached(imt_itree* t);
// all-time host milliseconds inside imt_itree_slice_prepare spent waiting for the plan set's previous slice to finish
// (back-pressure), as opposed to the step's own value check
double imt_itree_slice_backpressure_ms(const imt_itree* t);
