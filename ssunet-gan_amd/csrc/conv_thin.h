// Internal interface of the thin-conv kernels (conv_thin.hip); used by the conv/wgrad dispatchers.
#pragma once
#include "common.h"
int ssg_thin_conv_kind(const ssg_conv_desc* d);
int ssg_thin_conv_launch(const ssg_conv_desc* d, int kind, hipStream_t st);
// conv_wgrad4.hip: thin weight gradients on the 4x4x1 MFMA (kind 5 = dout thin, 6 = in thin)
int ssg_wgrad4_kind(const ssg_wgrad_desc* d);
int ssg_wgrad4_slices(const ssg_wgrad_desc* d, int kind, int* groups, long long* units_per_z, int* segs_per_row);
int ssg_wgrad4_launch(const ssg_wgrad_desc* d, int kind, hipStream_t st);
// conv_thin4.hip: thin convs on the 4x4x1 MFMA (kind 3 = in has 4 channels, 4 = Cout <= 4)
int ssg_thin4_conv_kind(const ssg_conv_desc* d);
int ssg_thin4_conv_launch(const ssg_conv_desc* d, int kind, hipStream_t st);
int ssg_thin4_conv_id(const ssg_conv_desc* d, int kind);      // profiling label id (12..15)
// conv_1x1.hip: 1x1 convs with 64 input channels as a streaming GEMM (weights in registers)
int ssg_conv1x1_k64_ok(const ssg_conv_desc* d);
int ssg_conv1x1_k64_launch(const ssg_conv_desc* d, hipStream_t st);
