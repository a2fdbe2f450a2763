// lds_asm.h -- LDS accesses hipcc must not see.  Behind an LDS-DMA (global_load_lds) into an LDS array the compiler puts
// `s_waitcnt vmcnt(0)` in front of every ds_read of that array, i.e. it drains the DMA queue before each fragment read and
// nothing overlaps.  Kernels that stream tiles with LDS-DMA while computing (wgrad.hip, tgemm.hip) therefore read LDS with
// inline asm and count their own lgkmcnt / vmcnt waits.
#pragma once
#include "mlp_device.h"

namespace mnr {

template <int OFF>
__device__ __forceinline__ float lds_ld(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// eight words OFF0, OFF0 + STRIDE, ... and the wait for them in ONE statement: the results are valid when the compiler sees them, so it
// may move or spill them freely (the split-precision weight-gradient path: its accumulators leave no register to spare, and a
// compiler-inserted copy of a register whose ds_read is still in flight would copy the old content)
template <int OFF0, int STRIDE>
__device__ __forceinline__ void lds_ld8_wait(unsigned addr, float (&v)[8]) {
    asm volatile("ds_read_b32 %0, %8 offset:%9\n\tds_read_b32 %1, %8 offset:%10\n\tds_read_b32 %2, %8 offset:%11\n\tds_read_b32 %3, %8 offset:%12\n\t"
                 "ds_read_b32 %4, %8 offset:%13\n\tds_read_b32 %5, %8 offset:%14\n\tds_read_b32 %6, %8 offset:%15\n\tds_read_b32 %7, %8 offset:%16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(addr), "n"(OFF0), "n"(OFF0 + STRIDE), "n"(OFF0 + 2 * STRIDE), "n"(OFF0 + 3 * STRIDE), "n"(OFF0 + 4 * STRIDE),
                   "n"(OFF0 + 5 * STRIDE), "n"(OFF0 + 6 * STRIDE), "n"(OFF0 + 7 * STRIDE)
                 : "memory");
}
__device__ __forceinline__ int lds_ld_i(unsigned addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// ... the same for a workgroup-uniform word: the value moves to an SGPR, so everything derived from it (job table
// indexing, tile addresses, loop control) stays on the scalar unit instead of VGPRs + vector loads from the kernel arguments
__device__ __forceinline__ int lds_ld_u(unsigned addr) { return __builtin_amdgcn_readfirstlane(lds_ld_i(addr)); }
__device__ __forceinline__ void lds_st_i(unsigned addr, int v) {
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(v) : "memory");
}
// (lds_addr, lds_ld4, wait_lgkm, pin: mlp_device.h -- the register-chained kernels read their A fragments the same way)
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// fetch-and-increment WITHOUT waiting for the result (hipcc's atomicAdd puts `s_waitcnt vmcnt(0)` right behind the
// instruction, which also drains the LDS-DMA queue); the value is valid after the caller's next wait_vm0()
__device__ __forceinline__ int atomic_inc_async(int32_t *p) {
    int v, one = 1;
    asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(v) : "v"(p), "v"(one) : "memory");
    return v;
}

}  // namespace mnr
