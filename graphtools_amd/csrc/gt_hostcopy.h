// Blocking copies between pageable host memory and the device (gt_hostcopy.cpp): on return the data has arrived.
// Both order themselves after the work already queued on ctx->stream.
#pragma once
#include <cstddef>

struct gt_ctx;

int gt_copy_to_host(gt_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int gt_copy_from_host(gt_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
// K values to the host with P = K / degree[row] derived by the copy lanes on the way (see gt_hostcopy.cpp)
int gt_fetch_kp_host(gt_ctx* ctx, double* K_host, double* P_host, const double* K_dev, long long nnz, const long long* indptr,
                     const double* degree, long long nrows, int* negative);
