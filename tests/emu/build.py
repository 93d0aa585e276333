"""Build librcmvs_emu.so: the product's kernel sources compiled for the CPU against tests/emu/hip/hip_runtime.h (see there).
TEST INFRASTRUCTURE ONLY.  The sources are copied to a scratch tree; the one textual change is the spelling of dynamic LDS
declarations (`extern __shared__ T name[];` has no host equivalent) -- everything else compiles as written."""
import glob
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CLANG = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
DYN_LDS = re.compile(r"extern\s+__shared__\s+((?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?)(\w+)\s+(\w+)\[\];")


def build(out_dir, only=None, verbose=False):
    """-> path of librcmvs_emu.so under out_dir.  only: iterable of .hip base names to include (default: all)."""
    src_root = os.path.join(out_dir, "src")
    csrc = os.path.join(src_root, "rc_mvsnet_amd", "csrc")
    shutil.rmtree(src_root, ignore_errors=True)
    os.makedirs(csrc)
    os.makedirs(os.path.join(src_root, "include"))
    shutil.copy(os.path.join(REPO, "include", "rcmvs.h"), os.path.join(src_root, "include", "rcmvs.h"))
    units = []
    for path in sorted(glob.glob(os.path.join(REPO, "rc_mvsnet_amd", "csrc", "*"))):
        name = os.path.basename(path)
        text = open(path).read()
        text = DYN_LDS.sub(lambda m: f"{m.group(2)}* {m.group(3)} = reinterpret_cast<{m.group(2)}*>(::shim::dyn_lds());", text)
        dst = os.path.join(csrc, name[:-4] + ".cpp" if name.endswith(".hip") else name)
        open(dst, "w").write(text)
        if name.endswith(".hip") and (only is None or name[:-4] in only):
            units.append(dst)
    lib = os.path.join(out_dir, "librcmvs_emu.so")
    flags = [CLANG, "-std=c++17", "-O1", "-g0", "-w", "-fPIC", "-ffp-contract=off", "-I", HERE]
    jobs = [(u, u[:-4] + ".o") for u in units] + [(os.path.join(HERE, "engine.cpp"), os.path.join(out_dir, "engine.o"))]

    def compile_one(job):
        subprocess.run(flags + ["-c", job[0], "-o", job[1]], check=True)
        return job[1]

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, jobs))
    subprocess.run([CLANG, "-shared", "-o", lib] + objs, check=True)
    if verbose:
        print("built", lib, "from", len(units), "kernel sources")
    return lib


def sources():
    return sorted(glob.glob(os.path.join(REPO, "rc_mvsnet_amd", "csrc", "*")) + glob.glob(os.path.join(HERE, "*.cpp")) +
                  glob.glob(os.path.join(HERE, "hip", "*.h")) + [os.path.join(REPO, "include", "rcmvs.h"), os.path.abspath(__file__)])


def build_cached(out_dir=None):
    """Build into tests/emu/_build (git-ignored) unless the library there is newer than every source."""
    out_dir = out_dir or os.path.join(HERE, "_build")
    lib = os.path.join(out_dir, "librcmvs_emu.so")
    def fresh():
        return os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(p) for p in sources())

    if fresh():
        return lib
    os.makedirs(out_dir, exist_ok=True)
    import fcntl
    with open(os.path.join(out_dir, ".lock"), "w") as lock:          # two test processes may find the library stale at the same moment
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return lib if fresh() else build(out_dir)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    import sys
    print(build(sys.argv[1] if len(sys.argv) > 1 else "/tmp/rcmvs_emu", verbose=True))
