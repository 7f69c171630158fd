"""Host-side mirror of `solve_merge_region` (reference src/merge_solver.rs:110-200).

The device does the expensive part — the all-pairs `optimize_sequences` exact-match test
(avk_optimize_pairs_batch); the classification on top of the pair matrix is a few set operations and
stays on the host, exactly as written in the reference.
"""
from ._abi import ZYG, RegionBatch


class MergeConfig:
    """MergeConfig (src/merge_solver.rs:62-84), same defaults"""

    def __init__(self, no_conflict_enabled=False, majority_voting_enabled=False, conflict_selection=None, max_branch_factor=50):
        self.no_conflict_enabled = no_conflict_enabled
        self.majority_voting_enabled = majority_voting_enabled
        self.conflict_selection = conflict_selection
        self.max_branch_factor = max_branch_factor


def pair_batch(multi_regions):
    """One CompareRegion-shaped item per (i < j) input pair of every MultiRegion
    (src/data_types/multi_region.rs): dicts {start, end, inputs:[variants...][, contig, region_id]}."""
    regions, owner = [], []
    for m, mr in enumerate(multi_regions):
        k = len(mr["inputs"])
        for i in range(k):
            for j in range(i + 1, k):
                regions.append({"start": mr["start"], "end": mr["end"], "contig": mr.get("contig", 0),
                                "truth": mr["inputs"][i], "query": mr["inputs"][j]})
                owner.append((m, i, j))
    return RegionBatch.from_regions(regions), owner


def solve_merge_regions(pairs_fn, multi_regions, config=None):
    """pairs_fn(batch, max_branch_factor) -> (status[], is_exact_match[]) is the device call
    (Context.optimize_pairs).  Returns one (status, classification) per region where classification is
    ("identical",) / ("no_conflict", indices) / ("majority", indices) / ("conflict_select", index) /
    ("different",), the reference's MergeClassification (src/data_types/merge_benchmark.rs:5-14)."""
    config = config or MergeConfig()
    batch, owner = pair_batch(multi_regions)
    status, exact = pairs_fn(batch, config.max_branch_factor) if batch.n_regions else ([], [])
    out = []
    per_region = {}
    for p, (m, i, j) in enumerate(owner):
        per_region.setdefault(m, []).append((i, j, int(status[p]), bool(exact[p])))
    for m, mr in enumerate(multi_regions):
        inputs = mr["inputs"]
        k = len(inputs)
        # variant_delta_length bails on an Unknown zygosity before anything else (:119-124, :216)
        if any((v[4] if len(v) > 4 else "Unknown") in ("Unknown", ZYG["Unknown"]) for vs in inputs for v in vs):
            out.append((6, None))
            continue
        err = 0
        all_identical, no_conflict = True, True
        match_sets = [{i} for i in range(k)]
        for (i, j, st, ex) in per_region.get(m, []):
            if st != 0:
                err = st
                break
            all_identical &= ex
            no_conflict &= (len(inputs[i]) == 0 or len(inputs[j]) == 0 or ex)
            if ex:
                match_sets[i].add(j)
                match_sets[j].add(i)
        if err:
            out.append((err, None))
            continue
        maj_count = k // 2 + 1
        first_maj = next((sorted(s) for s in match_sets if len(s) >= maj_count), [])
        if all_identical:
            cls = ("identical",)
        elif config.no_conflict_enabled and no_conflict:
            cls = ("no_conflict", [i for i, v in enumerate(inputs) if len(v)])
        elif config.majority_voting_enabled and first_maj:
            cls = ("majority", first_maj)
        elif config.conflict_selection is not None:
            cls = ("conflict_select", config.conflict_selection)
        else:
            cls = ("different",)
        out.append((0, cls))
    return out
