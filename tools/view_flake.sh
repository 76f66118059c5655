#!/bin/bash
# Repeats s5view's per-record and chunked pipelines on one small file and prints every run that fails (a stress loop for intermittent faults).
# bash tools/view_flake.sh [runs]
cd "$(dirname "$0")/.." || exit 1
N=${1:-200}
T=$(mktemp -d)
python - "$T" <<'P'
import struct, sys, numpy as np
sys.path.insert(0, ".")
from slow5tools_amd import press
rng = np.random.default_rng(21)
n = 700
sigs = [(480 + 35 * rng.standard_normal(int(k))).astype(np.int16) for k in rng.integers(50, 9000, n)]
sigs[5] = (480 + 35 * rng.standard_normal(120000)).astype(np.int16)
hdrs = [press.pack_hdr(b"r%06d" % i, i % 3, 8192.0, 23.0, 1467.61, 4000.0) for i in range(n)]
recs = press.encode_records(sigs, hdrs, None, press.REC_NONE, press.SIG_NONE)
text = b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
head = bytearray(64)
head[:6] = b"BLOW5\x01"; head[6:9] = bytes([0, 2, 0]); head[9] = 0; head[10:14] = struct.pack("<I", 3); head[14] = 0
open(sys.argv[1] + "/in.blow5", "wb").write(bytes(head) + struct.pack("<I", len(text)) + text + b"".join(recs) + b"5WOLB")
P
bad=0; notes=0
for i in $(seq 1 "$N"); do
    S5VIEW_PER_RECORD=1 slow5tools_amd/s5view "$T/in.blow5" "$T/a.blow5" zlib svb-zd 64 ${FLAKE_WORKERS:-2} 2> "$T/e1" || { bad=$((bad + 1)); echo "run $i per-record:"; grep -v "^:[0-9]:" "$T/e1" | tail -5; [ -n "$FLAKE_KEEP" ] && cp "$T/e1" "$FLAKE_KEEP/fail_$i.log"; }
    grep -h "not reached by device copies" "$T/e1" && notes=$((notes + 1))
    if [ -z "$FLAKE_PER_RECORD_ONLY" ]; then
        S5VIEW_CHUNK_KB=517 S5VIEW_READERS=3 slow5tools_amd/s5view "$T/in.blow5" "$T/b.blow5" zlib svb-zd 4096 2 2> "$T/e2" || { bad=$((bad + 1)); echo "run $i chunked:"; cat "$T/e2"; }
        cmp -s "$T/a.blow5" "$T/b.blow5" || { bad=$((bad + 1)); echo "run $i: outputs differ"; }
    fi
done
echo "view_flake: $bad bad of $N runs (x2 pipelines); $notes runs with an unreachable pinned buffer set aside (S5GPU_TRACE)"
rm -rf "$T"
