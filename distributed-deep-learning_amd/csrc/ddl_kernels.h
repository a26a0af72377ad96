// C launchers of the hand-written gfx950 kernels.  No torch headers here: every launcher takes
// raw device pointers and the hipStream_t to enqueue on, never allocates, never synchronises
// (so every launch sequence is hipGraph-capturable).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace ddl {

// ---- elementwise.hip -------------------------------------------------------------------
// uint8 NCHW [B,3,H,W] -> fp32 /255, NHWC ([B,H,W,3]) or NCHW.
void launch_normalize_u8(const uint8_t* in, float* out, int B, int C, int H, int W, bool nhwc,
                         hipStream_t s);
// fused softmax + cross-entropy (mean) + d(loss)/d(logits) + argmax; logits [B,C] row-major.
void launch_ce_fwd_bwd(const float* logits, const int64_t* targets, float* loss /*[1], zeroed here*/,
                       float* dlogits, int64_t* preds, int B, int C, hipStream_t s);
// flat Adam over [start,end) of four parallel arrays; step counter lives on the device.
void launch_adam_inc_step(float* step_t, float* aux /*[2]: 1-b1^t, 1/sqrt(1-b2^t)*/, double b1, double b2,
                          hipStream_t s);
void launch_adam_flat(float* p, float* g, float* m, float* v, const float* aux, int64_t start,
                      int64_t end, float lr, float b1, float b2, float eps, float wd, bool zero_grad,
                      hipStream_t s);

}  // namespace ddl
