"""DIAGNOSTICS ONLY.  The multi-frame reproducer (tools/mel_repro/mel_wave_multiframe.hip) as a hipGraph node beside the co-runners of
tests/parity_cases.case_mel_in_graph_beside_tails; with a -DMEL_DUMP build every bad replay's per-frame dump is compared word by word
with a good replay of the same input.   python tools/mel_repro/race.py tools/_melrepro_<name>.so [replays] [beside] [report.json]"""
import ctypes, json, sys
sys.path.insert(0, ".")
import torch
from desed_task_amd import _lib
from tests import parity_cases as P

so, replays = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 600
beside = sys.argv[3] if len(sys.argv) > 3 else "gemm"
report = sys.argv[4] if len(sys.argv) > 4 else None
_lib.use_library(None, is_emulator=False)
rep_lib = ctypes.CDLL(so)
DF = rep_lib.melrepro_dump_floats()
mel = P.make_mel()
mel._tables_to(torch.device("cuda"))
taps = torch.zeros(mel.TAPS_FLOATS, device="cuda")
_lib.get().call("sed_mel_taps", mel.fb_start.data_ptr(), mel.fb_len.data_ptr(), mel.fb_w.data_ptr(), mel.fb_stride, mel.n_mels,
                taps.data_ptr(), _lib.stream_ptr(taps))
torch.cuda.synchronize()
state = {"dbg": None}
P_ = ctypes.c_void_p
chk = torch.zeros(16 + 16 * 4000, dtype=torch.int32, device="cuda")
HAVE_CHK = rep_lib.melrepro_set_check(P_(chk.data_ptr())) == 0


def launch(audio, out):
    B, N = audio.shape
    T = 1 + N // mel.hop_length
    if DF and state["dbg"] is None:
        state["dbg"] = torch.zeros(B * T, DF, device="cuda")
    dbg = state["dbg"]
    rc = rep_lib.melrepro_fwd_wave(P_(audio.data_ptr()), P_(out.data_ptr()), B, N, T, mel.n_fft, mel.hop_length, mel.n_mels,
                                   P_(mel.window.data_ptr()), P_(mel.tw1024.data_ptr()), P_(mel.tw2048.data_ptr()),
                                   P_(mel.fb_start.data_ptr()), P_(mel.fb_len.data_ptr()), P_(mel.fb_w.data_ptr()), mel.fb_stride,
                                   P_(taps.data_ptr()), 0, P_(dbg.data_ptr() if dbg is not None else None),
                                   P_(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


good, findings, counts = {}, [], {"bad": 0}


def hw(word):
    return {"wave_slot": word & 15, "simd": (word >> 4) & 3, "pipe": (word >> 6) & 3, "cu": (word >> 8) & 15, "sh": (word >> 12) & 1,
            "se": (word >> 13) & 7, "queue": (word >> 24) & 7, "me": (word >> 30) & 3}


def after_replay(rep, i, nbad):
    if not DF:
        counts["bad"] += 1 if nbad else 0
        return
    d = state["dbg"]
    if not nbad:
        if i not in good:
            good[i] = d.clone()
        return
    counts["bad"] += 1
    if i not in good or len(findings) >= 40:
        return
    g = good[i]
    diff = (d[:, :5120].view(torch.int32) != g[:, :5120].view(torch.int32))
    for f in diff.any(1).nonzero().flatten().tolist():
        row = diff[f]
        sec = {"A_regs_after_pass3": row[:2048].view(16, 64, 2).any(2), "B_lds_readback": row[2048:4096].view(16, 64, 2).any(2),
               "C_magnitudes": row[4096:5120].view(16, 64)}
        info = d[f, 5120:5136].view(torch.int32).tolist()
        ginfo = g[f, 5120:5136].view(torch.int32).tolist()
        ent = {"replay": rep, "input": i, "frame": f, "frame_seq_in_wave": info[6], "workgroup": info[7], "wave": info[8],
               "hw": hw(info[0] & 0xFFFFFFFF), "xcc": info[1] & 15, "hw_good_replay": hw(ginfo[0] & 0xFFFFFFFF),
               "cycles": ((info[5] << 32) | (info[4] & 0xFFFFFFFF)) - ((info[3] << 32) | (info[2] & 0xFFFFFFFF)),
               "cycles_good_replay": ((ginfo[5] << 32) | (ginfo[4] & 0xFFFFFFFF)) - ((ginfo[3] << 32) | (ginfo[2] & 0xFFFFFFFF))}
        for name, m in sec.items():
            ent[name] = [{"q": int(q), "lanes": m[q].nonzero().flatten().tolist()} for q in m.any(1).nonzero().flatten().tolist()]
        # a sample of the differing words: value in the bad replay vs the good one
        a_bad = d[f, :2048].view(16, 64, 2); a_good = g[f, :2048].view(16, 64, 2)
        qs = sec["A_regs_after_pass3"].nonzero()[:3].tolist()
        ent["A_samples"] = [{"q": q, "lane": l, "bad": a_bad[q, l].tolist(), "good": a_good[q, l].tolist()} for q, l in qs]
        b_bad = d[f, 2048:4096].view(16, 64, 2); b_good = g[f, 2048:4096].view(16, 64, 2)
        qs = sec["B_lds_readback"].nonzero()[:3].tolist()
        ent["B_samples"] = [{"q": q, "lane": l, "bad": b_bad[q, l].tolist(), "good": b_good[q, l].tolist(),
                             "A_of_this_replay": a_bad[q, l].tolist()} for q, l in qs]
        findings.append(ent)


try:
    P.case_mel_in_graph_beside_tails("cuda", replays=replays, beside=beside, launch=launch, after_replay=after_replay)
    verdict = "bad 0"
except AssertionError as e:
    verdict = "FAILED " + str(e)[:300]
print(so, beside, "replays", replays, "bad replays", counts["bad"], "|", verdict[:200])
if HAVE_CHK:
    import struct
    torch.cuda.synchronize()
    c = chk.cpu().numpy().astype("uint32")
    n = int(c[0])
    f = lambda u: struct.unpack("f", struct.pack("I", int(u)))[0]
    events = []
    for i in range(min(n, 4000)):
        e = c[16 + 16 * i: 32 + 16 * i]
        ev = {"a": [f(e[0]), f(e[1])], "b": [f(e[2]), f(e[3])], "packed": [f(e[4]), f(e[5])], "scalar": [f(e[6]), f(e[7])],
              "packed_again": [f(e[8]), f(e[9])], "lane": int(e[10]) & 63, "wave": int(e[10]) >> 6, "workgroup": int(e[11]), "hw": hw(int(e[12])),
              "form": "a+ib (neg_lo)" if e[13] else "a-ib (neg_hi)", "t": int(e[14]),
              "lo_wrong": bool(e[4] != e[6]), "hi_wrong": bool(e[5] != e[7]), "again_right": bool(e[8] == e[6] and e[9] == e[7])}
        # what is the wrong half equal to?
        cands = {"a.x+b.y": ev["a"][0] + ev["b"][1], "a.x-b.y": ev["a"][0] - ev["b"][1], "a.x+b.x": ev["a"][0] + ev["b"][0], "a.x-b.x": ev["a"][0] - ev["b"][0],
                 "a.y+b.x": ev["a"][1] + ev["b"][0], "a.y-b.x": ev["a"][1] - ev["b"][0], "a.y+b.y": ev["a"][1] + ev["b"][1], "a.y-b.y": ev["a"][1] - ev["b"][1],
                 "a.x": ev["a"][0], "a.y": ev["a"][1], "b.x": ev["b"][0], "b.y": ev["b"][1]}
        import numpy as np
        for half, idx in (("lo", 0), ("hi", 1)):
            if ev[half + "_wrong"]:
                ev[half + "_equals"] = [k for k, v_ in cands.items() if np.float32(v_) == np.float32(ev["packed"][idx])]
        events.append(ev)
    import collections
    print("MEL_CHECK: %d mismatching +-i packed adds;" % n, "lanes", sorted(collections.Counter(e["lane"] // 16 for e in events).items()),
          "forms", collections.Counter(e["form"] for e in events).most_common(), "lo/hi wrong", collections.Counter((e["lo_wrong"], e["hi_wrong"]) for e in events).most_common(),
          "repeat right", collections.Counter(e["again_right"] for e in events).most_common(),
          "lo equals", collections.Counter(tuple(e.get("lo_equals", ())) for e in events).most_common(5),
          "hi equals", collections.Counter(tuple(e.get("hi_equals", ())) for e in events).most_common(5))
    # events per (workgroup, wave, time): how many lanes per event group
    groups = collections.Counter((e["workgroup"], e["wave"], e["t"] >> 8) for e in events)
    print("events per (workgroup, wave, 256-cycle window):", collections.Counter(groups.values()).most_common(6))
    for ev in events[:5]:
        print(json.dumps(ev)[:700])
    findings.append({"check_events": events[:400], "check_count": n})
if report:
    json.dump({"library": so, "beside": beside, "replays": replays, "bad_replays": counts["bad"], "findings": findings}, open(report, "w"), indent=1)
for ent in findings[:6]:
    print(json.dumps(ent)[:1500])
