"""README.md's and DESIGN.md's numbers are GENERATED from one evidence visit's kept files (tools/make_tables.py) -- round 5's tables
quoted a visit whose kernel sources were not the tree's.  Here: the visit named in profiles/round6/FINAL was made on the tree's
kernel sources (same fingerprint), its line and its detail file are of one run, and both documents carry exactly the block the kept
files give."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import bench_core  # noqa: E402
import make_tables  # noqa: E402


def test_the_kept_evidence_is_of_the_trees_kernel_sources():
    tag = make_tables.final_tag()
    fp = make_tables.fingerprint(tag)
    assert fp == bench_core.kernel_fingerprint(), (
        f"profiles/round6 holds visit {tag} of kernel sources {fp}, the tree's are {bench_core.kernel_fingerprint()}: "
        "run tools/visit_full.sh + tools/keep_visit.sh + tools/make_tables.py --write again")
    line = json.load(open(os.path.join(make_tables.RDIR, f"bench_{tag}.json")))
    detail = json.load(open(os.path.join(make_tables.RDIR, f"bench_detail_{tag}.json")))
    assert line["config"]["kernel_sources_sha256_16"] == fp
    assert (detail["value"], detail["ms_per_step"]) == (line["value"], line["ms_per_step"]), "line and detail file are of two runs"
    assert line["roofline"]["traffic"], "the kept line carries its PMC traffic"
    traffic = json.load(open(os.path.join(make_tables.RDIR, "pmc_traffic.json")))
    assert traffic["kernel_sources_sha256_16"] == fp


def test_documents_quote_exactly_what_the_kept_files_give():
    block = make_tables.block()
    for doc in ("README.md", "DESIGN.md"):
        text = open(os.path.join(ROOT, doc)).read()
        assert make_tables.replace_block(text, block) == text, f"{doc}: run python tools/make_tables.py --write"
    # the headline's three figures recomputed from the kept files, by hand
    tag = make_tables.final_tag()
    fp = make_tables.fingerprint(tag)
    line = json.load(open(os.path.join(make_tables.RDIR, f"bench_{tag}.json")))
    algo = 64 * (4 * 3840 * 2160 + 65536)
    assert abs(algo / (line["ms_per_step"] * 1e-3) / 1e9 / 8000.0 - line["roofline"]["frac"]) < 2e-4
    ns, calls = make_tables.rocprof_avg_ns(tag, fp, "headline_only")
    assert calls > 1000 and abs(algo / (ns * 1e-9) / 1e9 / 8000.0 - line["roofline"]["frac_kernel"]) < 0.04, \
        "rocprofv3's average of the headline kernel agrees with the line's own stamps (another process, another pool: a few per cent)"
    # the SQ counters are the accumulation kernel's, not those of the K0 launches that prepared the frames in the same process
    valu = make_tables.sq(fp, "three_scope_x64_uniform", "SQ_INSTS_VALU")
    assert valu is not None and 15.0 < valu * 64 / (64 * 3840 * 2160) < 40.0, "SQ_INSTS_VALU per texel of the three-scope kernel"
    assert f"`{fp}`" in block and f"**{line['value']:.0f}**" in block and f"**{line['roofline']['frac']:.3f}**" in block


def test_no_stale_fingerprints_outside_history_notes():
    fp = make_tables.fingerprint(make_tables.final_tag())
    for doc in ("README.md", "DESIGN.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for old in ("f5e7d2dc216d273f", "a2118265a2afc592", "57263b4608419f98"):
            assert old not in text or old == fp, f"{doc} still quotes the sources fingerprint {old} of an earlier round"
