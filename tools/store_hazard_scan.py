"""Scan the device assembly of csrc/*.hip for the store-data hazard of DESIGN.md section 4 ("A store-data hazard the compiler does not
know"): a store of more than 8 bytes whose data registers are written by a VALU instruction one or two instructions later with no
s_nop in between.  Runs on the build machine (no GPU):   python tools/store_hazard_scan.py
hipcc 7.2 leaves that pattern behind buffer stores whose scalar offset is a register; global / flat stores get their wait state."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gan-image-captioning_amd", "csrc")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    lines = [l.strip() for l in open(path) if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    hits = []
    for i, l in enumerate(lines):
        if not re.match(r"(buffer|global|flat)_store_dwordx[34]", l):
            continue
        ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
        data = regs(ops[0]) if l.startswith("buffer") else regs(ops[1])
        for k in (1, 2):
            if i + k >= len(lines):
                break
            n = lines[i + k]
            if n.startswith("s_nop"):
                break
            if n.startswith("v_") and regs(n.split(None, 1)[1].split(",")[0].strip()) & data:
                hits.append((l, k, n))
    return hits


def main(names=None):
    """names: source basenames to scan (default: every csrc/*.hip).  Returns the number of distance-1 sites."""
    bad = 0
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) if not names else [os.path.join(CSRC, n) for n in names]
    with tempfile.TemporaryDirectory() as tmp:
        for src in srcs:
            out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
                            "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
            hits = scan(out)
            # one instruction of distance is the hazard; two is what the compiler leaves behind global stores (its own rule: one wait state)
            near = [h for h in hits if h[1] == 1]
            print(f"{os.path.basename(src):24s} distance 1: {len(near)}   distance 2: {len(hits) - len(near)}")
            for l, k, n in near:
                print("   ", l, "|", n)
            bad += len(near)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1:]) else 0)
