"""Provenance of the committed profile summaries bench.py reads.

Every json under profiles/ whose numbers end up in the bench line carries
    "provenance": {"source_sha": sha256 of the kernel sources + the host package, "bench_sha": sha256 of bench.py, "commit": ..., "dirty": ...}
`source_sha` is computed on the GPU box right after the profile is collected (the box has the source tree, not .git):
    python scripts/provenance.py stamp gpurun_out/x/hbm_by_kernel.json ...
`commit` is added in the build container when the summary is copied into profiles/:
    python scripts/provenance.py commit profiles/r5_*.json
bench.py recomputes `source_sha` at run time (source_digest) and reports a profile's traffic only when it matches -- otherwise
`traffic: null, traffic_stale: true`: a kernel changed after the counters were collected."""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCE_GLOBS = ("transception_amd/csrc/*.hip", "transception_amd/csrc/*.h", "transception_amd/csrc/*.inc", "transception_amd/csrc/*.py",
                "transception_amd/*.py", "include/*.h")


def _sha(paths, root):
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(os.path.relpath(p, root).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()


def source_digest(root: str = ROOT) -> str:
    """sha256 over every kernel source (incl. the generated attention streams and their generators), the C-ABI header and the host package."""
    files = [f for g in SOURCE_GLOBS for f in glob.glob(os.path.join(root, g))]
    return _sha(files, root)


def bench_digest(root: str = ROOT) -> str:
    return _sha([os.path.join(root, "bench.py"), os.path.join(root, "scripts", "bench_stage.py")], root)


def matches(doc: dict, root: str = ROOT) -> bool:
    """True when `doc` (a loaded profile summary) was collected from exactly the sources under `root`."""
    p = doc.get("provenance") if isinstance(doc, dict) else None
    return bool(p) and p.get("source_sha") == source_digest(root)


def _git(*a):
    try:
        return subprocess.run(["git", *a], cwd=ROOT, capture_output=True, text=True, check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        return None


def main():
    mode, files = sys.argv[1], sys.argv[2:]
    for f in files:
        doc = json.load(open(f))
        if not isinstance(doc, dict):
            continue
        p = doc.setdefault("provenance", {})
        if mode == "stamp":
            p["source_sha"], p["bench_sha"] = source_digest(), bench_digest()
        elif mode == "commit":
            p["commit"] = _git("rev-parse", "HEAD")
            p["dirty"] = bool(_git("status", "--porcelain", "--", "transception_amd", "include", "bench.py"))
            p["source_sha_at_commit_time"] = source_digest()
        json.dump(doc, open(f, "w"), indent=1)
        print(f, p)


if __name__ == "__main__":
    main()
