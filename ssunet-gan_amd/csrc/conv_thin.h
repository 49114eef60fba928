// Internal interface of the thin-conv kernels (conv_thin.hip); used by the conv/wgrad dispatchers.
#pragma once
#include "common.h"
int ssg_thin_conv_kind(const ssg_conv_desc* d);
int ssg_thin_conv_launch(const ssg_conv_desc* d, int kind, hipStream_t st);
int ssg_thin_wgrad_kind(const ssg_wgrad_desc* d);
int ssg_thin_wgrad_splits(const ssg_wgrad_desc* d, long long* pix_per_block);
int ssg_thin_wgrad_launch(const ssg_wgrad_desc* d, int kind, hipStream_t st);
