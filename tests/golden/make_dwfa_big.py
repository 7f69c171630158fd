#!/usr/bin/env python3
"""Extracts the two DNA test strings of the reference's `test_big_early_termination`
(src/dwfa/dynamic_wfa.rs:453-468) into dwfa_big.json.  Only the two string literals (test
data) are taken; run once in the dev container where /root/reference exists.  The committed
JSON is what the tests read."""
import json, re, sys
src = open("/root/reference/src/dwfa/dynamic_wfa.rs").read()
c1 = re.search(r'let c1 =\s+"([ACGT]+)";', src).group(1)
s23 = re.search(r'let seq_23 = "([ACGT]+)";', src).group(1)
json.dump({
    "source": "src/dwfa/dynamic_wfa.rs:453-468 test_big_early_termination",
    "baseline": s23, "other": c1,
    "max_ed_during_updates": 2, "ed_after_updates": 2, "ed_after_finalize": 5278,
}, open(sys.argv[1] if len(sys.argv) > 1 else "dwfa_big.json", "w"), indent=1)
print(len(c1), len(s23))
