// hopperrender_amd/csrc/hf_async_io.hip -- pinned asynchronous host I/O of a context on side streams (include/hopperflow.h
// hf_update_frame_async / hf_download_frame_async, hf_wait_flow / hf_wait_download): the reference's blocking CL_TRUE transfers
// (opticalFlowCalcSDR.cpp:19-42) as H2D / D2H copies that overlap the compute stream.  Layout of the ABI: hf_ctx.h.

#include "hf_ctx.h"

using namespace hfi;

namespace hfi {

int io_init(hf_ctx* c) {
    if (c->io_in) return HF_OK;
    HF_HIP(c, hipStreamCreateWithFlags(&c->io_in, hipStreamNonBlocking));
    HF_HIP(c, hipStreamCreateWithFlags(&c->io_out, hipStreamNonBlocking));
    HF_HIP(c, hipEventCreateWithFlags(&c->ev_h2d, hipEventDisableTiming));
    HF_HIP(c, hipEventCreateWithFlags(&c->ev_last_launch, hipEventDisableTiming));
    HF_HIP(c, hipEventCreateWithFlags(&c->ev_out_ready, hipEventDisableTiming));
    for (auto& e : c->ev_slot_prep) HF_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->ev_d2h) HF_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->ev_dl) HF_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->out_ring[0] = c->out_frame;
    for (int i = 1; i < hf_ctx::kOutRing; i++) HF_HIP(c, hipMalloc(&c->out_ring[i], c->out_bytes));
    return HF_OK;
}

// Before a warp/copy writes into an output-ring slot: wait for the asynchronous readback that still uses it.
int guard_output_slot(hf_ctx* c, const void* target, hipStream_t launch_stream) {
    if (!c->io_out) return HF_OK;
    for (int i = 0; i < hf_ctx::kOutRing; i++)
        if (target == c->out_ring[i] && c->d2h_pending[i]) {
            HF_HIP(c, hipStreamWaitEvent(launch_stream, c->ev_d2h[i], 0));
            c->d2h_pending[i] = false;
        }
    return HF_OK;
}

int note_launch(hf_ctx* c, hipStream_t launch_stream) {   // remembers "the frames/flow of the ring are being read up to here"
    if (!c->io_in) return HF_OK;
    HF_HIP(c, hipEventRecord(c->ev_last_launch, launch_stream));
    c->have_last_launch = true;
    return HF_OK;
}

}  // namespace hfi

extern "C" {

int hf_update_frame_async(hf_ctx* c, const void* pinned_host_frame) {
    HF_CHECK_CTX(c);
    if (!pinned_host_frame) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_update_frame_async: null frame");
    if (c->batch) return fail(c, HF_ERR_STATE, "hf_update_frame_async: the context is a member of a batch (asynchronous host I/O uses side streams of its own)");
    if (int rc = set_device(c)) return rc;
    if (int rc = io_init(c)) return rc;
    // the slot about to be overwritten holds the oldest frame: its last readers are the warp/copy launches
    // issued so far and its own (three updates old) phase-plane build
    if (c->have_last_launch) HF_HIP(c, hipStreamWaitEvent(c->io_in, c->ev_last_launch, 0));
    HF_HIP(c, hipStreamWaitEvent(c->io_in, c->ev_slot_prep[0], 0));
    HF_HIP(c, hipMemcpyAsync(c->ring_store[0], pinned_host_frame, c->in_bytes, hipMemcpyHostToDevice, c->io_in));
    HF_HIP(c, hipEventRecord(c->ev_h2d, c->io_in));
    if (int rc = leave_warp_stream(c)) return rc;
    if (c->timing()) {
        HF_HIP(c, hipEventRecord(c->ev_upload, c->stream));
        c->upload_recorded = true;
    }
    HF_HIP(c, hipStreamWaitEvent(c->stream, c->ev_h2d, 0));
    c->ring[0] = c->ring_store[0];
    hf::launch_prep_frame(c->g, c->pl, c->ring[0], c->pp[0], c->stream);
    c->plane_pending[0] = false;
    HF_HIP(c, hipGetLastError());
    HF_HIP(c, hipEventRecord(c->ev_slot_prep[0], c->stream));
    rotate_after_upload(c);
    return HF_OK;
}

int hf_download_frame_async(hf_ctx* c, void* pinned_host_out) {
    HF_CHECK_CTX(c);
    if (!pinned_host_out) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_download_frame_async: null buffer");
    if (c->batch) return fail(c, HF_ERR_STATE, "hf_download_frame_async: the context is a member of a batch (asynchronous host I/O uses side streams of its own)");
    if (int rc = set_device(c)) return rc;
    if (int rc = io_init(c)) return rc;
    hipStream_t last = c->on_warp_stream ? c->warp_stream : c->stream;   // where the frame was just produced
    HF_HIP(c, hipEventRecord(c->ev_out_ready, last));
    HF_HIP(c, hipStreamWaitEvent(c->io_out, c->ev_out_ready, 0));
    HF_HIP(c, hipMemcpyAsync(pinned_host_out, c->out_target, c->out_bytes, hipMemcpyDeviceToHost, c->io_out));
    HF_HIP(c, hipEventRecord(c->ev_dl[c->dl_issued % hf_ctx::kDlRing], c->io_out));
    c->dl_issued++;
    for (int i = 0; i < hf_ctx::kOutRing; i++)
        if (c->out_target == c->out_ring[i]) {   // internal output: the next frame goes to the next ring slot
            HF_HIP(c, hipEventRecord(c->ev_d2h[i], c->io_out));
            c->d2h_pending[i] = true;
            c->out_idx = (i + 1) % hf_ctx::kOutRing;
            c->out_target = c->out_ring[c->out_idx];
            break;
        }
    c->warp_started = false;
    return HF_OK;
}

int hf_wait_flow(hf_ctx* c) {
    HF_CHECK_CTX(c);
    if (int rc = set_device(c)) return rc;
    if (!c->async()) return HF_OK;                       // blocking contexts have finished every call already
    if (c->batch) return fail(c, HF_ERR_STATE, "hf_wait_flow: the context is a member of a batch (use hf_sync)");
    if (!c->flow_done_recorded) return HF_OK;
    HF_HIP(c, hipEventSynchronize(c->ev_flow_done));
    if (c->delta_pending) { c->total_frame_delta = *c->h_total_delta; c->delta_pending = false; }
    return HF_OK;
}

uint64_t hf_downloads_issued(const hf_ctx* c) { return c ? c->dl_issued : 0; }

int hf_wait_download(hf_ctx* c, uint64_t index) {
    HF_CHECK_CTX(c);
    if (index >= c->dl_issued) return fail(c, HF_ERR_INVALID_ARGUMENT, "hf_wait_download: download %llu has not been issued (%llu so far)",
                                           (unsigned long long)index, (unsigned long long)c->dl_issued);
    if (int rc = set_device(c)) return rc;
    // The ring keeps the events of the last kDlRing downloads.  A slot that has been reused belongs to a LATER download of the same
    // in-order stream, so waiting for it (or, for an index older than the whole ring, for the oldest event still kept) implies that
    // download `index` has landed.
    const uint64_t oldest = c->dl_issued > (uint64_t)hf_ctx::kDlRing ? c->dl_issued - (uint64_t)hf_ctx::kDlRing : 0;
    const uint64_t wait_for = index < oldest ? oldest : index;
    HF_HIP(c, hipEventSynchronize(c->ev_dl[wait_for % hf_ctx::kDlRing]));
    return HF_OK;
}

}  // extern "C"
