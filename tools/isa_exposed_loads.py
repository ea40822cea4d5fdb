"""ISA audit: global loads whose latency is exposed.  Walks every kernel's instruction stream in program order, keeps the
outstanding vector-memory operations (gfx9: loads AND stores count in vmcnt) and, at every `s_waitcnt vmcnt(N)`, reports the loads
that this wait retires together with how much work was issued between the load and the wait (instructions / MFMAs).
A load inside a loop that is waited for after < MIN_DIST instructions has its whole memory latency on the critical path -- typically
a load the compiler left inside a divergent branch (the value is needed at the merge point, or is used inside the branch), or a
register loaded before the loop whose conservative wait drains the loop's own prefetch (see sed_pin in csrc/sed_common.h).
Usage: python tools/isa_exposed_loads.py [-d MIN_DIST] [file.hip ...]   (default: every csrc/*.hip, MIN_DIST 12).
tests/test_isa_audit.py keeps the kernels of the default training step clean."""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "desed_task_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def audit_kernel(insts, min_dist):
    """-> {(mnemonic, distance, mfmas in between): count} of loads inside loops retired after < min_dist instructions."""
    labels = {l[:-1]: i for i, l in enumerate(insts) if l.endswith(":")}
    inloop = [False] * len(insts)
    for i, l in enumerate(insts):
        m = re.match(r"s_c?branch\w*\s+(\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            for j in range(labels[m.group(1)], i + 1):
                inloop[j] = True
    out, rows = [], {}          # outstanding: (index, is_load, mnemonic)
    for i, l in enumerate(insts):
        op = l.split()[0]
        if op.startswith(("global_load", "buffer_load", "global_store", "buffer_store", "global_atomic")):
            out.append((i, "load" in op, op))
        elif op == "s_waitcnt" and "vmcnt(" in l:
            n = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
            retire, out = (out[:len(out) - n], out[len(out) - n:]) if n < len(out) else ([], out)
            for (j, is_load, mn) in retire:
                if not is_load or not inloop[j]:
                    continue
                body = [x for x in insts[j + 1:i] if not x.endswith(":")]
                if len(body) < min_dist:
                    key = (mn, len(body), sum(x.startswith("v_mfma") for x in body))
                    rows[key] = rows.get(key, 0) + 1
    return rows


def compile_asm(path):
    """gfx950 ISA text of one .hip source, compiled with the product's flags."""
    return subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", CSRC,
                           "-I", os.path.join(ROOT, "include"), path, "-o", "-"], capture_output=True, text=True).stdout


def audit_file(path, min_dist=12, seen=None, asm_text=None):
    """-> [(kernel, mnemonic, distance, mfmas, count)] for one .hip source (compiled to gfx950 ISA with the product's flags, or taken
    from `asm_text`).  `seen` (a set) collects the names of all kernels found in the file."""
    asm = (asm_text if asm_text is not None else compile_asm(path)).splitlines()
    found, kern, insts = [], None, []
    for line in asm:
        s = line.split(";")[0].strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            kern = re.sub(r"^void ", "", kern).split("(")[0]
            insts = []
            if seen is not None:
                seen.add(kern)
            continue
        if kern and s.startswith(".Lfunc_end"):
            for (mn, dist, mf), cnt in sorted(audit_kernel(insts, min_dist).items(), key=lambda kv: kv[0][1]):
                found.append((kern, mn, dist, mf, cnt))
            kern = None
            continue
        if not kern or not s or s.startswith("."):
            if kern and re.match(r"^\.LBB\w+:", s):
                insts.append(s)
            continue
        insts.append(s)
    return found


if __name__ == "__main__":
    args = sys.argv[1:]
    min_dist = 12
    if args[:1] == ["-d"]:
        min_dist = int(args[1]); args = args[2:]
    for f in args or sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        for kern, mn, dist, mf, cnt in audit_file(f, min_dist):
            print("%-24s %-64s %2d x %-22s waited after %2d instructions (%d MFMA)" % (os.path.basename(f), kern[:64], cnt, mn, dist, mf))
