// imt_sliced_transport.hpp -- the imt_transport handle shared by imt_sliced.cpp and imt_sliced_rccl.cpp.
#pragma once
#include <memory>
#include <string>
#include "imt_ctx.hpp"
#include "imt_sliced_sched.hpp"

struct imt_transport {
    std::unique_ptr<imt::sliced::Transport> impl;
    imt_ctx* ctx = nullptr;          // whose last_error carries the transport's messages (null: custom / local)
    std::string error;
    int users = 0;
    bool abandoned = false;          // the last world on it never drained (imt_sliced_destroy): destroy leaks the device side
};

// takes ownership of t (deleted if the handle cannot be allocated)
imt_transport* imt_transport_wrap(imt::sliced::Transport* t, imt_ctx* ctx);
