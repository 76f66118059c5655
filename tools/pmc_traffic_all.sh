#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes: each fills the TCC counter budget) of the dominant kernel of every
# bench leg, written to profiles/pmc_traffic.json with the hash of the kernel sources it was collected on (bench.py only quotes a figure
# whose hash matches the build it runs).  hbm_bytes_per_read = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / reads: the factor 2 is the
# gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE tallies 128-byte requests at 64 bytes).  Run on the GPU box.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/pmc_traffic}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name counter command...
  local name=$1 c=$2; shift 2
  rm -rf /tmp/pt_${name}_$c
  ( cd $R && timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pt_${name}_$c -o pt -- "$@" ) > $OUT/${name}_$c.log 2>&1
}
for c in FETCH_SIZE WRITE_SIZE; do
  run enc $c python tools/stream_time.py 400000
  run svb $c python bench.py --svb-only --reads 400000 --steps 3 --warmup 1 --cpu-seconds 0
  run svbs $c python tools/svb_stream_time.py 400000
  run long $c python bench.py --long --long-reads 16384 --steps 2 --warmup 1 --min-leg-steps-long 2 --cpu-seconds 0
  run decnp $c python tools/decode_bulk.py 1000000 4000 np 3
  run decfull $c python tools/decode_bulk.py 1000000 4000 full 3
  run mixed $c python bench.py --mixed --steps 2 --warmup 1 --cpu-seconds 0
  run dec4k $c python bench.py --decode --decode-batches-only --reads 200000 --cpu-seconds 0
done
# instruction issue of the headline encoder and the decoder (one more pass each): vector instructions per read — the ceiling that binds those kernels
run enc SQ_INSTS_VALU python tools/stream_time.py 400000
run decnp SQ_INSTS_VALU python tools/decode_bulk.py 1000000 4000 np 3
python3 - <<PY
import csv, glob, json, sys, os
sys.path.insert(0, "$R")
import bench
sha = bench.csrc_sha256()
def per_launch(name, counter, kernels):
    f = glob.glob("/tmp/pt_%s_%s/**/*counter_collection.csv" % (name, counter), recursive=True)[0]
    tot = {}
    for r in csv.DictReader(open(f)):
        for k in kernels:
            if k in r["Kernel_Name"] and r["Counter_Name"] == counter:
                tot.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in tot.items()}, {k: len(v) for k, v in tot.items()}
legs = [("enc", "k_encode_stream", ["k_encode_stream"], 400000, 4000),
        ("svb", "k_svbzd_encode", ["k_svbzd_encode"], 400000, 4000),
        ("svbs", "k_svbzd_stream", ["k_svbzd_stream"], 400000, 4000),
        ("long", "k_pack+k_deflate_staged", ["k_pack", "k_deflate_staged"], 4096, 100000),   # 16384 reads = 4 chunks of 4096 per step
        ("decnp", "k_inflate_par_np", ["k_inflate_par_np"], 1000000, 4000),
        ("decfull", "k_inflate_par+k_unpack", ["k_inflate_par<1"], 1000000, 4000),
        ("mixed", "k_encode_fused+k_pack+k_deflate_staged+k_compact", ["k_encode_fused", "k_pack", "k_deflate_staged", "k_compact"], 262144, "mixed"),   # one launch of each per step
        ("dec4k", "k_inflate_par_np@K", ["k_inflate_par_np"], 4096, 4000)]      # get batches of K = 4096 records (the last batch of a pass is shorter: a few % low)
out = []
for name, label, kernels, reads, n in legs:
    try:
        fe, nf = per_launch(name, "FETCH_SIZE", kernels)
        wr, nw = per_launch(name, "WRITE_SIZE", kernels)
        fetch = sum(fe.values()); write = sum(wr.values())
        b = (2 * fetch + write) * 1024 / reads
        print("%-26s %8d reads x %6s: FETCH %14.1f KiB  WRITE %14.1f KiB per launch (%s launches) -> %.1f B/read" % (label, reads, n, fetch, write, nf, b))
        ent = {"kernel": label, "samples_per_read": n, "hbm_bytes_per_read": round(b, 1), "fetch_KiB_per_launch": round(fetch, 1), "write_KiB_per_launch": round(write, 1),
               "reads_per_launch": reads, "source": "tools/pmc_traffic_all.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH x 2: gfx950 correction)",
               "csrc_sha256": sha}
        if name in ("enc", "decnp"):
            try:
                va, nv = per_launch(name, "SQ_INSTS_VALU", kernels)
                ent["valu_insts_per_read"] = round(sum(va.values()) / reads, 1)      # wave-level vector instructions per read (SQ_INSTS_VALU, its own pass)
                print("%-26s SQ_INSTS_VALU %.4g per launch -> %.1f per read" % (label, sum(va.values()), ent["valu_insts_per_read"]))
            except Exception as e:
                print(label, "SQ_INSTS_VALU failed:", repr(e))
        out.append(ent)
    except Exception as e:
        print(label, "failed:", repr(e))
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
print("csrc_sha256", sha)
PY
