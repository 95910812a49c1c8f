// HIP kernels of the GKR sumcheck hot path for gfx950 (MI355X).  Included once by vpgpu.hip.
//
// All tables are arrays of F (16 B, {real,img}); lane i of a wave touches element base+i, so every wave
// instruction is one contiguous 1 KiB global_load/store_dwordx4.  Tables carry a `valid` length: entries
// at or beyond it are zero by convention and are neither stored nor loaded (SHA-256 layer sizes are not
// powers of two, so this trims the padding the reference keeps, src/prover.cpp:472-482).
//
// Unlike the reference, which stores every bookkeeping entry as a linear polynomial {a,b} (32 B) and
// folds it lazily (src/prover.cpp:483-485), the device keeps only folded VALUES (16 B): the table of
// round k is T_k[g] = T_{k-1}[2g] + r_{k-1}*(T_{k-1}[2g+1] - T_{k-1}[2g]), and the round polynomial is
// assembled from pairs (T_k[2p], T_k[2p+1]).  Same field values, half the bytes.
//
// Files: vp_kernels_round.h (reductions, eq tables, evaluation, per-round kernels)  vp_kernels_batch.h (batched init +
// fold kernels)  vp_kernels_plan.h (segment / closing kernels, batched launches)  vp_kernels_pc.h (polynomial commitment)  vp_kernels_fftgkr.h (circuit and tables of fft_gkr).
#pragma once
#include <hip/hip_runtime.h>
#include "vp_field.h"
#include "vp_check.h"
#include "vp_kernels_round.h"
#include "vp_kernels_persist.h"
#include "vp_kernels_batch.h"
#include "vp_kernels_plan.h"
#include "vp_kernels_pc.h"
#include "vp_kernels_fftgkr.h"
