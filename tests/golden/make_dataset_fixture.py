"""Derive the small dataset fixture from the reference's annotation file (run in the build container only):

    python tests/golden/make_dataset_fixture.py

Reads /root/reference/all.json (data, not code) and stores ONLY what pins ClipPairDataset's behaviour
(/root/reference/CLIP/train.py:63-91): per key, the class labels in first-seen order with their counts, plus
caption byte-length statistics.  No annotation text other than the class labels themselves is copied."""
import collections
import json
import os

SRC = "/root/reference/all.json"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "all_json_summary.json")


def main():
    data = json.load(open(SRC))
    ann = data["annotations"]
    out = {"n_annotations": len(ann), "keys": {}}
    for key in ("violation_type", "caption_type"):
        kept = [a for a in ann if a[key] != ""]
        c = collections.Counter(a[key] for a in kept)          # insertion order = first-seen order, as the reference relies on
        out["keys"][key] = {"labels": list(c.keys()), "counts": list(c.values()), "n_nonempty": len(kept)}
    for key in ("violation_list", "caption"):
        lens = [len(a[key].encode("utf-8")) for a in ann if a[key] != ""]
        out["keys"][key] = {"n_nonempty": len(lens), "max_utf8_bytes": max(lens), "n_over_75_bytes": sum(l > 75 for l in lens)}
    json.dump(out, open(OUT, "w"), ensure_ascii=False, indent=1)
    print(json.dumps(out, ensure_ascii=False)[:600])


if __name__ == "__main__":
    main()
