// Canonical fp32 parameter block of the PointNet cloud encoder (dpf_hip.h: dpf_encoder_canon_floats), shared by the
// eval-mode kernel (encoder.hip) and the training-mode kernels (encoder_train.hip).
#ifndef DPF_ENCODER_LAYOUT_H
#define DPF_ENCODER_LAYOUT_H

namespace {

constexpr int EC0 = 3, EC1 = 64, EC2 = 128, EC3 = 256, EC4 = 512;
// per layer l = 0..3: W[cout][cin], gamma[cout], beta[cout], running_mean[cout], running_var[cout]
__host__ __device__ constexpr int e_layer_off(int l) {
    return l == 0 ? 0 : l == 1 ? EC1 * EC0 + 4 * EC1 : l == 2 ? EC1 * EC0 + 4 * EC1 + EC2 * EC1 + 4 * EC2
                                                              : EC1 * EC0 + 4 * EC1 + EC2 * EC1 + 4 * EC2 + EC3 * EC2 + 4 * EC3;
}
__host__ __device__ constexpr int e_cin(int l) { return l == 0 ? EC0 : l == 1 ? EC1 : l == 2 ? EC2 : EC3; }
__host__ __device__ constexpr int e_cout(int l) { return l == 0 ? EC1 : l == 1 ? EC2 : l == 2 ? EC3 : EC4; }
constexpr int E_CANON = e_layer_off(3) + EC4 * EC3 + 4 * EC4;      // 176 064 floats

}  // namespace
#endif
