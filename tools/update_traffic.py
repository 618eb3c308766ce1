#!/usr/bin/env python
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) of `python bench.py ...` -> profiles/hbm_traffic.json.

usage: tools/update_traffic.py <workload key> <dir with FETCH_SIZE/ and WRITE_SIZE/ pass subdirs> [source note]
HBM bytes per launch of every kernel = 2 x FETCH_SIZE + WRITE_SIZE (counter unit KiB; the factor 2 is the gfx950
correction for wide coalesced reads, /opt/skills/guides/MI355X_MICROARCH.md section HBM), averaged over the launches of
the run.  The file carries the sha of the kernel sources it was measured on; bench.py quotes it only when that matches."""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
key, root = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
SEG = {"td_prepare": "td_prepare+td_mark_docs", "td_mark_docs": "td_prepare+td_mark_docs", "td_prepare_mark": "td_prepare+td_mark_docs",
       "td_far_probe": "td_split_tiles", "td_tail": "td_merge_pieces", "td_giant_scan": "td_long_pieces+td_giant_pieces+td_scan_tiles",
       "td_split_tiles": "td_split_tiles", "td_split_far_pieces": "td_split_tiles", "td_split_far_tiles": "td_split_tiles",
       "td_probe_tiles": "td_probe_tiles", "td_merge_pieces": "td_merge_pieces",  # (segments = the events of TD_OPT_PROFILE)
       "td_long_pieces": "td_long_pieces+td_giant_pieces+td_scan_tiles", "td_giant_pieces": "td_long_pieces+td_giant_pieces+td_scan_tiles",
       "td_scan_tiles": "td_long_pieces+td_giant_pieces+td_scan_tiles", "td_pack_tokens": "td_pack_tokens",
       "td_pack_plain": "td_pack_tokens", "td_pack_rest": "td_pack_tokens", "td_collect_misses": "td_merge_pieces", "td_copy_dups": "td_merge_pieces"}
def per_kernel(counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(root, counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("td::", "")
            acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: acc[k] / cnt[k] * 1024.0 for k in acc}
fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
kern = {k: int(2 * fetch.get(k, 0) + write.get(k, 0)) for k in set(fetch) | set(write) if k.startswith("td_")}
ent = {"per_kernel": kern}
for k, v in kern.items():
    if k in SEG:
        ent[SEG[k]] = ent.get(SEG[k], 0) + v
ent["_all"] = sum(kern.values())
if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):  # the bench line of the measured command: the launch's algorithmic bytes
    try:
        j = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
        c = j["config"]
        ent["algorithmic_bytes"] = int(c["bytes_rank0"] + 4 * c["tokens_rank0"] + 8 * (c["docs_rank0"] + 1))
        ent["ratio_to_algorithmic"] = round(ent["_all"] / ent["algorithmic_bytes"], 3)
        ent["per_kernel_ratio"] = {k: round(v / ent["algorithmic_bytes"], 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]) if v >= 0.005 * ent["_all"]}
    except Exception as e:  # noqa: BLE001
        ent["algorithmic_bytes_error"] = str(e)
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
doc = json.load(open(path)) if os.path.exists(path) else {}
sha = bench.kernel_source_sha()
if doc.get("kernel_source_sha") != sha:
    doc = {"kernel_source_sha": sha}
doc["source"] = note or doc.get("source", "")
doc[key] = ent
json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
print(key, json.dumps(ent))
