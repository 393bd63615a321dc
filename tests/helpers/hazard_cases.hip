// Deliberately broken (bad_*) and repaired (good_*) kernels for
// tools/isa_hazards.py: tests/test_cabi_host.py compiles this file for gfx950
// and expects every bad_* kernel to be reported and no good_* kernel.
// Never linked into the library, never run.
#include <hip/hip_runtime.h>

// A. an untracked look-ahead load whose hand-counted wait is one too high:
// the second load may still be in flight when its register is read
extern "C" __global__ void bad_wait_count(const double *a, const double *b, double *out)
{
    double x, y;
    const unsigned off = threadIdx.x * 8u;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(x) : "v"(off), "s"(a));
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(y) : "v"(off), "s"(b));
    asm volatile("s_waitcnt vmcnt(1)" : "+v"(x), "+v"(y));
    out[threadIdx.x] = x + y;
}

extern "C" __global__ void good_wait_count(const double *a, const double *b, double *out)
{
    double x, y;
    const unsigned off = threadIdx.x * 8u;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(x) : "v"(off), "s"(a));
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(y) : "v"(off), "s"(b));
    asm volatile("s_waitcnt vmcnt(1)" : "+v"(x));
    double s = x * 2.0;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(y));
    out[threadIdx.x] = s + y;
}

// A'. an exit from the middle of a look-ahead loop: on the path that leaves
// early the register is read without any wait
extern "C" __global__ void bad_loop_exit(const double *a, double *out, int n)
{
    double x = 0.0, acc = 0.0;
    unsigned off = threadIdx.x * 8u;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(x) : "v"(off), "s"(a));
    for (int i = 0; i < n; i++) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x));
        acc += x;
        off += 512u;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(x) : "v"(off), "s"(a));
    }
    out[threadIdx.x] = acc + x;       // x: the last look-ahead, never waited for
}

// B. VALU write of a VGPR, then v_readfirstlane of it with no wait state
// (the compiler's hazard recognizer does not look inside inline asm)
extern "C" __global__ void bad_readfirstlane(const int *in, int *out)
{
    int s, t;
    asm volatile("v_add_u32 %1, %2, %2\n\tv_readfirstlane_b32 %0, %1"
                 : "=s"(s), "=&v"(t) : "v"(in[threadIdx.x]));
    out[threadIdx.x] = s;
}

extern "C" __global__ void good_readfirstlane(const int *in, int *out)
{
    int s, t;
    asm volatile("v_add_u32 %1, %2, %2\n\ts_nop 0\n\tv_readfirstlane_b32 %0, %1"
                 : "=s"(s), "=&v"(t) : "v"(in[threadIdx.x]));
    out[threadIdx.x] = s;
}

// C. VALU write of a VGPR, then a DPP read of it one instruction later
extern "C" __global__ void bad_dpp(const int *in, int *out)
{
    int t, r;
    asm volatile("v_add_u32 %0, %2, %2\n\t"
                 "s_nop 0\n\t"
                 "v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf"
                 : "=&v"(t), "=&v"(r) : "v"(in[threadIdx.x]));
    out[threadIdx.x] = r;
}

extern "C" __global__ void good_dpp(const int *in, int *out)
{
    int t, r;
    asm volatile("v_add_u32 %0, %2, %2\n\t"
                 "s_nop 1\n\t"
                 "v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf"
                 : "=&v"(t), "=&v"(r) : "v"(in[threadIdx.x]));
    out[threadIdx.x] = r;
}

// D. VALU write of an SGPR pair, then a memory instruction that reads it as
// its scalar base fewer than five wait states later
extern "C" __global__ void bad_sgpr_base(const int *in, int lo, int hi, int *out)
{
    int v;
    const unsigned off = threadIdx.x * 4u;
    int vlo = lo + (int)threadIdx.x * 0, vhi = hi + (int)threadIdx.x * 0;
    asm volatile("v_readfirstlane_b32 s20, %1\n\t"
                 "v_readfirstlane_b32 s21, %2\n\t"
                 "s_nop 1\n\t"
                 "global_load_dword %0, %3, s[20:21]\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v) : "v"(vlo), "v"(vhi), "v"(off) : "s20", "s21", "memory");
    out[threadIdx.x] = v;
}

extern "C" __global__ void good_sgpr_base(const int *in, int lo, int hi, int *out)
{
    int v;
    const unsigned off = threadIdx.x * 4u;
    int vlo = lo + (int)threadIdx.x * 0, vhi = hi + (int)threadIdx.x * 0;
    asm volatile("v_readfirstlane_b32 s20, %1\n\t"
                 "v_readfirstlane_b32 s21, %2\n\t"
                 "s_nop 4\n\t"
                 "global_load_dword %0, %3, s[20:21]\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v) : "v"(vlo), "v"(vhi), "v"(off) : "s20", "s21", "memory");
    out[threadIdx.x] = v;
}
