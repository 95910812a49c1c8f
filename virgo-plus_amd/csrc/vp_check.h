// Device-side checked build (-DVP_CHECKED; SURVEY.md §5 "a debug build with index asserts in the scatter / gather kernels").  The product build compiles
// every VP_CHK away.  A checked build keeps the bounds of the circuit uploaded LAST in one device-global descriptor (a checked process proves one
// circuit at a time) and records the FIRST violated check — site number and three values — in a device-global record that every entry point reads back
// when its stream has drained (vp_check_collect in vpgpu.hip): the call then fails with VP_EHIP and "device check failed: site N (...)".  A failed check
// never traps (a trapping wave can take the device down, see the pool's rules): the guarded access is skipped or redirected to element 0.
//   sites:  1 operand gather of an init contribution (layer, index)      2 V gather of a phase-2 slot (layer, index)      3 row pointers of a target-sorted list
//           4 Liu gather list (row pointers, half-table selector)        5 store of a folded table entry beyond the buffers' capacity
//           6 LDS slot of a transform tile                               7 codeword position of a leaf
#pragma once

namespace vp {

#ifdef VP_CHECKED
struct VpChkDesc { unsigned n_layers; unsigned lsize[64]; unsigned long long table_cap; unsigned n_liu_tables_max; };
__device__ VpChkDesc g_vp_chk;
__device__ unsigned int g_vp_chk_err[4];
__device__ __forceinline__ bool vp_chk_fail(unsigned site, unsigned a, unsigned b, unsigned c) {
    if (atomicCAS(&g_vp_chk_err[0], 0u, site) == 0u) { g_vp_chk_err[1] = a; g_vp_chk_err[2] = b; g_vp_chk_err[3] = c; }
    return false;
}
#define VP_CHK(cond, site, a, b, c) ((cond) ? true : vp::vp_chk_fail((site), (unsigned) (a), (unsigned) (b), (unsigned) (c)))
#define VP_CHK_LAYER(l, x) VP_CHK((unsigned) (l) < vp::g_vp_chk.n_layers && (unsigned) (x) < vp::g_vp_chk.lsize[(unsigned) (l) & 63u], 1, (l), (x), vp::g_vp_chk.lsize[(unsigned) (l) & 63u])
__device__ __forceinline__ unsigned g_vp_chk_layers() { return g_vp_chk.n_layers; }
__device__ __forceinline__ unsigned g_vp_chk_lsize(unsigned l) { return g_vp_chk.lsize[l & 63u]; }
__device__ __forceinline__ unsigned g_vp_chk_liu() { return g_vp_chk.n_liu_tables_max; }
__device__ __forceinline__ unsigned long long g_vp_chk_cap() { return g_vp_chk.table_cap; }
#else
#define VP_CHK(cond, site, a, b, c) (true)
#define VP_CHK_LAYER(l, x) (true)
#endif

}  // namespace vp
