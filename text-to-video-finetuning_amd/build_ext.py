"""Build libt2v_hip.so (gfx950) in-tree with hipcc.  `python build_ext.py [--force]`."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libt2v_hip.so")
SOURCES = ["gemm.hip", "gemm_w8.hip", "norm.hip", "attn.hip", "elementwise.hip", "lora_wgrad.hip", "lora_merge.hip", "temporal_fused.hip"]
# translation units: (source, object, extra flags).  gemm_w8.hip is compiled three times with -DW8_PART=0/1/2 (plain kernels + entry
# points / LR = 1 kernels / LR = 2 kernels): its ~60 kernel instantiations took 3 minutes in one hipcc process
UNITS = [(s, s.replace(".hip", ".o"), []) for s in SOURCES if s != "gemm_w8.hip"] + \
        [("gemm_w8.hip", f"gemm_w8_p{i}.o", [f"-DW8_PART={i}"]) for i in range(3)]


def _digest():
    h = hashlib.sha256()
    # every source AND every header under csrc/ (colsum.h is included by both GEMM files: an edit there must not ship a
    # stale library behind a matching stamp)
    headers = sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))
    for f in SOURCES + headers:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    with open(os.path.join(INCLUDE, "t2v_abi.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    stamp = LIB + ".sha256"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    # per-object stamps (source + every header): an edit to one .hip recompiles that object only
    hh = hashlib.sha256()
    for f in sorted(f for f in os.listdir(CSRC) if f.endswith(".h")):
        with open(os.path.join(CSRC, f), "rb") as fh:
            hh.update(f.encode())
            hh.update(fh.read())
    with open(os.path.join(INCLUDE, "t2v_abi.h"), "rb") as fh:
        hh.update(fh.read())
    stamps = {}
    for src, oname, flags in UNITS:
        obj = os.path.join(CSRC, oname)
        objs.append(obj)
        h = hh.copy()
        h.update(" ".join(flags).encode())
        with open(os.path.join(CSRC, src), "rb") as fh:
            h.update(fh.read())
        stamps[obj] = h.hexdigest()
        ostamp = obj + ".sha256"
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == stamps[obj]:
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC,
               "-Wno-unused-result"] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, obj, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src} failed ---\n{out}\n")
        else:
            if verbose and out.strip():          # warnings: shown, and the object is stamped all the same (it did compile)
                print(out)
            with open(obj + ".sha256", "w") as f:
                f.write(stamps[obj])
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
