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
