"""DIAGNOSTICS ONLY: packed-fp32 instruction forms (tools/mel_repro/pk_probe.hip) executed millions of times while the split-bf16 GEMM
(MFMA waves) runs on the same CUs -- in a hipGraph fork, in eager launches on two streams, and alone.  python tools/mel_repro/pk_probe.py"""
import ctypes, sys
sys.path.insert(0, ".")
import torch
from desed_task_amd import _lib
_lib.use_library(None, is_emulator=False)
lib = _lib.get()
so = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else "tools/_pkprobe.so")
P_ = ctypes.c_void_p
NF = 18
FORMS = ["pk_add op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]  (a + i b)", "pk_add op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]  (a - i b)", "pk_add (no op_sel)",
         "pk_add op_sel:[0,1] op_sel_hi:[1,0]", "pk_add op_sel:[1,0] op_sel_hi:[0,1]", "pk_mul op_sel:[1,1] op_sel_hi:[1,0]",
         "pk_fma op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]", "pk_fma op_sel:[1,0,0]", "pk_fma op_sel:[0,1,0]", "pk_add op_sel_hi:[1,0]",
         "pk_add op_sel:[0,1]", "pk_mul (no op_sel)", "pk_add op_sel:[0,1] op_sel_hi:[1,0] between s_nop 7 x 2", "pk_add SAME SUM, swapped operand as src0: op_sel:[1,0] op_sel_hi:[0,1]",
         "pk_add op_sel:[0,1] op_sel_hi:[1,0] + dependent pk_add", "pk_add op_sel:[0,1] op_sel_hi:[1,0] under s_setprio 3", "pk_fma op_sel:[0,0,1]", "pk_mul op_sel:[0,1]"]
grid, iters = 512, 4000
n = grid * 256
g = torch.Generator().manual_seed(11)
inp = torch.randn(8 * n, generator=g).cuda()
counts = torch.zeros(NF * 8 + 8 + 8 * 64, dtype=torch.int32, device="cuda")
H, Ms = 128, 48 * 156
h_big = torch.randn(Ms, H, device="cuda"); gi = torch.zeros(Ms, 2, 3 * H, device="cuda")
w0 = torch.randn(3 * H, H, device="cuda") * 0.1; w1 = torch.randn(3 * H, H, device="cuda") * 0.1; b0 = torch.zeros(3 * H, device="cuda")


CO = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
big = torch.randn(2048, 2048, device="cuda")
go = torch.empty(48, 156, 2 * H, device="cuda"); whh = torch.randn(3 * H, H, device="cuda") * 0.1; gi6 = torch.randn(48, 156, 2, 3 * H, device="cuda")


def storm(stream, k):
    if CO == "f32mfma":         # the exact-fp32 GEMM (v_mfma_f32_32x32x2f32)
        for _ in range(k // 4):
            lib.call("sed_gemm_pair", h_big.data_ptr(), h_big.data_ptr(), w0.data_ptr(), w1.data_ptr(), b0.data_ptr(), b0.data_ptr(),
                     gi.data_ptr(), gi.data_ptr() + 3 * H * 4, Ms, 3 * H, H, H, H, 6 * H, 0, 1, 1, 0, stream.cuda_stream)
        return
    if CO == "rocblas":
        with torch.cuda.stream(stream):
            for _ in range(k // 8):
                big @ big
        return
    if CO == "gru":             # no MFMA: the BiGRU recurrence (VALU + LDS)
        for _ in range(k // 8):
            lib.call("sed_gru_fwd", gi6.data_ptr(), whh.data_ptr(), whh.data_ptr(), b0.data_ptr(), b0.data_ptr(), go.data_ptr(), None, 48, 156, H, stream.cuda_stream)
        return
    for _ in range(k):
        lib.call("sed_gemm_pair_bf16x3", h_big.data_ptr(), h_big.data_ptr(), w0.data_ptr(), w1.data_ptr(), b0.data_ptr(), b0.data_ptr(),
                 gi.data_ptr(), gi.data_ptr() + 3 * H * 4, Ms, 3 * H, H, H, H, 6 * H, 0, 1, 1, 0, stream.cuda_stream)


def probe(stream):
    assert so.pk_probe(P_(inp.data_ptr()), P_(counts.data_ptr()), iters, n, grid, P_(stream.cuda_stream)) == 0


def report(title):
    torch.cuda.synchronize()
    c = counts.cpu().numpy().astype("int64")
    total = grid * 256 * iters
    print("== %s: %.2e executions of every form per lane-quarter" % (title, total / 4))
    for f in range(NF):
        row = c[f * 8:(f + 1) * 8].reshape(4, 2)
        if row.sum():
            print("   %-62s wrong lo / hi per quarter (lanes 0-15 .. 48-63): %s" % (FORMS[f], row.tolist()))
    if not c[:NF * 8].sum():
        print("   no mismatch in any form")
    counts.zero_()


main, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(main):
    probe(main)
report("probe alone")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(main):
    e0.record(); probe(main); e1.record()
torch.cuda.synchronize(); print("probe kernel: %.2f ms" % e0.elapsed_time(e1)); counts.zero_()
with torch.cuda.stream(main):
    e0.record(); storm(main, 10); e1.record()
torch.cuda.synchronize(); per = e0.elapsed_time(e1) / 10; print("GEMM launch: %.1f us" % (per * 1e3))
for rep in range(20):
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s2):
        storm(s2, 200)
    with torch.cuda.stream(s1):
        probe(s1)
    main.wait_stream(s1); main.wait_stream(s2)
report("eager, two streams (20 rounds), co-runner " + CO)


def body():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        storm(s2, 200)
    s1.wait_stream(cur)
    with torch.cuda.stream(s1):
        probe(s1)
    cur.wait_stream(s1); cur.wait_stream(s2)


with torch.cuda.stream(main):
    body()
torch.cuda.synchronize(); counts.zero_()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=main, capture_error_mode="thread_local"):
    body()
for rep in range(20):
    graph.replay()
report("hipGraph fork { probe | co-runner %s } (20 replays)" % CO)
