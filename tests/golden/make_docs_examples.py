"""Extracts the worked examples of the reference's user documentation (docs/compare.md, docs/merge.md of PacificBiosciences/aardvark
v0.10.5) into tests/golden/docs_examples.json: the example rows of summary.tsv, of the labeled VCF, of the two debug tables and of the
merged VCF.  They are the only byte-level known answers the reference gives for its writers (its source tree has no tests for
src/writers/).  Run in the build container, where /root/reference exists:  python tests/golden/make_docs_examples.py"""
import json
import os

REF = "/root/reference/docs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "docs_examples.json")


def code_blocks(path):
    blocks, cur = [], None
    for line in open(path).read().split("\n"):
        if line.startswith("```"):
            if cur is None:
                cur = []
            else:
                blocks.append(cur)
                cur = None
        elif cur is not None:
            cur.append(line)
    return [b for b in blocks if b]


def block_starting(blocks, prefix):
    for b in blocks:
        if b[0].startswith(prefix):
            return [l for l in b if l != "..."]
    raise KeyError(prefix)


compare = code_blocks(os.path.join(REF, "compare.md"))
merge = code_blocks(os.path.join(REF, "merge.md"))
out = {
    "source": "PacificBiosciences/aardvark docs/compare.md and docs/merge.md (v0.10.5): example rows as printed there",
    "summary_tsv": block_starting(compare, "compare_label\tcomparison"),
    "labeled_vcf": block_starting(compare, "#CHROM\tPOS"),
    "region_sequences": block_starting(compare, "region_id\tcoordinates\tref_seq"),
    "region_summary": block_starting(compare, "region_id\tcoordinates\tcomparison"),
    "merged_vcf": block_starting(merge, "#CHROM\tPOS"),
    "merge_summary": block_starting(merge, "merge_reason\tvariant_type"),
}
json.dump(out, open(OUT, "w"), indent=1)
print({k: len(v) for k, v in out.items() if isinstance(v, list)})
