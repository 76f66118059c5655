"""Copy the end-of-round profile set (gpurun_out/r03p, made by tools/run_r03_profiles.sh on the GPU box; r02p / run_r02_profiles.sh in round 2) into profiles/ under
round-tagged names and rewrite profiles/pmc_traffic.json (HBM bytes per read from the PMC passes, stamped with the hash of the
kernel sources they were collected on: bench.py prints roofline.traffic only while the hash still matches).
python tools/install_profiles.py [src_dir] [tag]"""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

src = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r02p")
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
P = os.path.join(ROOT, "profiles")
for d in sorted(os.listdir(src)):
    p = os.path.join(src, d)
    if os.path.isdir(p):
        if os.path.exists(os.path.join(p, "kernel_stats.csv")):
            shutil.copy(os.path.join(p, "kernel_stats.csv"), os.path.join(P, "%s_kernel_stats_%s.csv" % (tag, d)))
        if os.path.exists(os.path.join(p, "stdout.txt")):
            shutil.copy(os.path.join(p, "stdout.txt"), os.path.join(P, "%s_stdout_%s.txt" % (tag, d)))
    elif d.endswith(".json"):
        shutil.copy(p, os.path.join(P, "%s_bench_line_%s.json.txt" % (tag, d[:-5].replace("bench_", ""))))
    elif d.endswith(".txt"):
        shutil.copy(p, os.path.join(P, "%s_%s" % (tag, d)))
# traffic: tools/pmc_traffic_all.sh (round 3 on) writes the JSON itself, one entry per leg's dominant kernel, stamped with the source hash
pj = os.path.join(src, "pmc", "pmc_traffic.json")
if os.path.exists(pj):
    entries = json.load(open(pj))
    stale = [e["kernel"] for e in entries if e.get("csrc_sha256") != bench.csrc_sha256()]
    if stale:
        print("WARNING: kernel sources changed since the PMC pass; bench.py will not quote traffic for", stale)
    for e in entries:
        e["source"] = "profiles/%s_pmc_traffic.txt; " % tag + e.get("source", "")
    json.dump(entries, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    for e in entries: print(e["kernel"], e["samples_per_read"], e["hbm_bytes_per_read"], e["csrc_sha256"])
    sys.exit(0)
# traffic: 2 x FETCH_SIZE + WRITE_SIZE (KiB per launch; the x 2 is the gfx950 correction of MI355X_MICROARCH.md) / reads per launch
txt = open(os.path.join(src, "pmc_k_encode_stream.txt")).read()
reads = int(re.search(r"over (\d+) reads", txt).group(1))
fetch = float(re.search(r"FETCH_SIZE\s+per-launch avg\s+([\d.]+)", txt).group(1))
write = float(re.search(r"WRITE_SIZE\s+per-launch avg\s+([\d.]+)", txt).group(1))
entry = {"kernel": "k_encode_stream", "samples_per_read": 4000, "hbm_bytes_per_read": round((2 * fetch + write) * 1024 / reads, 1),
         "source": "profiles/%s_pmc_k_encode_stream.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB per launch of %d reads; FETCH x 2: gfx950 correction)" % (tag, reads),
         "csrc_sha256": bench.csrc_sha256()}
entries = [entry]
dt = os.path.join(src, "pmc_decode_traffic.txt")
if os.path.exists(dt):      # bulk decode: the inflating wave also unpacks, so one kernel's traffic is the whole decode's
    t = open(dt).read()
    f2 = re.search(r"FETCH_SIZE\s+per-launch\s+([\d.]+)\s+grid (\d+)", t)
    w2 = re.search(r"WRITE_SIZE\s+per-launch\s+([\d.]+)", t)
    if f2 and w2:
        recs = int(f2.group(2)) // 64
        entries.append({"kernel": "k_inflate_par+k_unpack", "samples_per_read": 4000,
                        "hbm_bytes_per_read": round((2 * float(f2.group(1)) + float(w2.group(1))) * 1024 / recs, 1),
                        "source": "profiles/%s_pmc_decode_traffic.txt (k_inflate_par<1> over %d records; FETCH x 2: gfx950 correction)" % (tag, recs),
                        "csrc_sha256": bench.csrc_sha256()})
json.dump(entries, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
for e in entries: print(e)
