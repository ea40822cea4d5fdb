"""VALU issue-rate probe (diagnostics): 32-bit multiply-add vs 24-bit multiply vs shift/xor/add vs v_rcp_f32, 8 independent
chains per thread, 8 waves per SIMD.  Measured on MI355X: 914 / 521 / 921 / 1738 us -> v_mad_u32_u24 is full rate, the 32-bit
multiply-add costs two issue slots (not four), so the three 32-bit multiplies of sed_hash are ~6 of its ~14 slots: a 24-bit
variant would save ~3 slots per dropout decision -- not worth changing the mask definition for."""
import ctypes, os, subprocess, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = r"""#include <hip/hip_runtime.h>
#include <stdint.h>
template <int MODE>
__global__ void k(uint32_t* out, int iters, uint32_t s) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7919u + i * 104729u + s;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = a[i] * 0x9E3779B1u + 12345u;                 // 32-bit multiply-add
            else if (MODE == 1) a[i] = __umul24(a[i], 0xD1B54Bu) + 12345u;     // 24-bit multiply
            else if (MODE == 2) a[i] = (a[i] ^ (a[i] >> 15)) + 12345u;          // shift-xor-add
            else { float f = __uint_as_float((a[i] & 0x7FFFFFu) | 0x3F800000u); f = __builtin_amdgcn_rcpf(f); a[i] = __float_as_uint(f) + 12345u; }
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
extern "C" void run(int mode, uint32_t* out, int iters, void* stream) {
    dim3 g(256 * 8), b(256);
    if (mode == 0) hipLaunchKernelGGL(k<0>, g, b, 0, (hipStream_t)stream, out, iters, 1u);
    if (mode == 1) hipLaunchKernelGGL(k<1>, g, b, 0, (hipStream_t)stream, out, iters, 1u);
    if (mode == 2) hipLaunchKernelGGL(k<2>, g, b, 0, (hipStream_t)stream, out, iters, 1u);
    if (mode == 3) hipLaunchKernelGGL(k<3>, g, b, 0, (hipStream_t)stream, out, iters, 1u);
}
"""
so = os.path.join(HERE, "_imul_probe.so")
if not os.path.exists(so):
    src = os.path.join(HERE, "_imul_probe.hip")
    open(src, "w").write(SRC)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", src, "-o", so])
    os.remove(src)
if not torch.cuda.is_available():
    sys.exit(0)
lib = ctypes.CDLL(so)
out = torch.empty(256 * 8 * 256, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mode, name in enumerate(["a*K+c (u32)", "umul24(a,K)+c", "(a^(a>>15))+c", "rcp + add"]):
    for _ in range(2):
        lib.run(mode, ctypes.c_void_p(out.data_ptr()), 4096, ctypes.c_void_p(st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.run(mode, ctypes.c_void_p(out.data_ptr()), 4096, ctypes.c_void_p(st)); e1.record(); torch.cuda.synchronize()
    print("%-16s %.1f us" % (name, e0.elapsed_time(e1) * 1e3))
