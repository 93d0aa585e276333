#!/usr/bin/env python
"""Static instruction mix of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only).
   python tools/dev/isa_stats.py file.s [substring-of-kernel-name ...]
Counts per kernel: VALU (packed / plain / transcendental), SALU, LDS, vector memory loads / stores, waitcnts, branches, and the
VGPR / SGPR / scratch figures of the metadata.  Straight-line kernels only make sense here (unrolled loops): no trip counts."""
import re, sys, collections
path = sys.argv[1]
filt = sys.argv[2:]
txt = open(path).read().split("\n")
kern = None
stats = collections.OrderedDict()
for ln in txt:
    m = re.match(r"^(_Z\w+):\s", ln)
    if m:
        kern = m.group(1); stats[kern] = collections.Counter(); continue
    if kern is None: continue
    if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
        kern = None; continue
    s = ln.strip()
    if not s or s.startswith(";") or s.startswith("."): continue
    op = s.split()[0]
    c = stats[kern]
    if op.startswith("v_pk_"): c["valu_pk"] += 1
    elif op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")): c["valu_trans"] += 1
    elif op.startswith("v_mfma"): c["mfma"] += 1
    elif op.startswith("v_"): c["valu"] += 1
    elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
    elif op.startswith("s_barrier"): c["barrier"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): c["branch"] += 1
    elif op.startswith("s_"): c["salu"] += 1
    elif op.startswith("ds_"): c["lds_" + ("w" if ("write" in op or "store" in op) else "r")] += 1
    elif op.startswith(("buffer_load", "global_load", "flat_load")): c["vmem_ld"] += 1
    elif op.startswith(("buffer_store", "global_store", "flat_store")): c["vmem_st"] += 1
    elif op.startswith(("scratch_",)): c["scratch"] += 1
    else: c["other"] += 1
meta = {}
cur = None
for ln in txt:
    m = re.match(r"\s+\.name:\s+(\S+)", ln)
    if m: cur = m.group(1); meta[cur] = {}
    for key in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count"):
        m = re.match(r"\s+\." + key + r":\s+(\d+)", ln)
        if m and cur: meta[cur][key] = int(m.group(1))
import subprocess
for k, c in stats.items():
    try: dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception: dem = k
    if filt and not any(f in dem for f in filt): continue
    md = meta.get(k, {})
    print(f"{dem}\n    vgpr {md.get('vgpr_count')} sgpr {md.get('sgpr_count')} scratch {md.get('private_segment_fixed_size')} | " +
          " ".join(f"{a}={b}" for a, b in sorted(c.items())))
