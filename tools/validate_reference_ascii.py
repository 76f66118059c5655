#!/usr/bin/env python3
"""Run in the build container only (needs /root/reference): every record line of every .slow5 file the reference ships must survive
line -> payload -> line through the ORACLE unchanged.  Pins the text conventions (double/float printing, ".", enum indices, arrays)
on all of the reference's own ASCII data, not just the three ASCII/binary twins committed under tests/golden/."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind as ob

base = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/test/data"
files = lines = bad_files = 0
kinds = set()
fails = []
for dp, _, fs in os.walk(base):
    for f in sorted(fs):
        if not f.endswith(".slow5"):
            continue
        p = os.path.join(dp, f)
        raw = open(p, "rb").read().split(b"\n")
        try:
            k = next(i for i, l in enumerate(raw) if l.startswith(b"#read_id"))
            types = ob.aux_types(raw[k - 1])
        except (StopIteration, AssertionError):
            bad_files += 1          # deliberately malformed inputs of the reference's error tests
            continue
        files += 1
        kinds.update(types)
        for l in raw[k + 1:]:
            if not l:
                continue
            lines += 1
            pay = ob.line_to_payload(l + b"\n", types)
            back = ob.payload_to_line(pay, types) if pay else None
            if back != l + b"\n":
                fails.append((os.path.relpath(p, base), l[:60], (back or b"")[:60]))
print("files %d (skipped %d without a parsable header), record lines %d, aux type codes seen %s" % (files, bad_files, lines, sorted(kinds)))
print("round-trip failures: %d" % len(fails))
seen = set()
for f in fails:
    if f[0] not in seen:
        seen.add(f[0]); print("  ", f)
