"""Adversarial random cases (tests/randcase.py): the read-centric core built for the host vs the oracle here, the HIP
kernels vs the oracle on the GPU box."""
import numpy as np
import pytest

import randcase
from hostsim import sim

SEEDS = list(range(60))


def _oracle(oracle_lib, arr, rs, stranded, combine):
    return oracle_lib.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos,
                                rs.pos, rs.flag, rs.cig_off, rs.cigar, stranded, combine)


@pytest.mark.parametrize("odd", [False, True])
@pytest.mark.parametrize("stranded", [0, 1, 2])
def test_host_core_matches_oracle_on_random_cases(stranded, odd, oracle_lib):
    """``odd``: BED strands that are none ('?', '.') in stranded analyses too and junctions whose ends coincide -- tables with several
    sites at one position and partner lists that name a position twice (tools/fuzz_reference.py --odd-strands holds the table and
    the oracle to the reference itself on these)."""
    hits = 0
    for seed in SEEDS:
        arr, rs = randcase.make_case(seed, bool(stranded), odd=odd)
        for combine in (0, 1):
            want = _oracle(oracle_lib, arr, rs, stranded, combine)
            got = sim.count(arr, rs, stranded, combine)
            for w, g in zip(want, got):
                assert np.array_equal(w, g), (seed, combine)
            hits += int(want[2].sum())
    assert hits > 0   # double counts do occur: the rival paths are exercised


@pytest.mark.gpu
@pytest.mark.parametrize("odd", [False, True])
@pytest.mark.parametrize("stranded", [0, 1, 2])
@pytest.mark.parametrize("kernel", ["ranges", "pairs", "ranges_agg"])
def test_gpu_matches_oracle_on_random_cases(stranded, kernel, odd, oracle_lib):
    from spliser_amd import native
    flags = {"ranges": 0, "pairs": native.OPT_PAIR_KERNEL, "ranges_agg": native.OPT_WAVE_AGGREGATION}[kernel]
    with native.Context(0) as ctx:
        for seed in SEEDS:
            arr, rs = randcase.make_case(seed, bool(stranded), odd=odd)
            if arr.n == 0:
                continue
            s = native.SiteArrays.from_chrom(arr)
            r = native.ReadArrays(rs.pos, rs.flag, rs.cig_off, rs.cigar)
            for combine in (0, 1):
                want = _oracle(oracle_lib, arr, rs, stranded, combine)
                got = ctx.count(s, r, stranded, combine, flags)
                for w, g in zip(want, got):
                    assert np.array_equal(w, g), (seed, combine, kernel)
                for cryptic in (False, True):
                    ws = oracle_lib.beta2_sse(arr.pos, arr.part_off, arr.part_pos, arr.part_site, arr.alpha, arr.edge_cnt, *want, cryptic)
                    gs = ctx.sse(s, got[0], got[1], got[2], cryptic)
                    for w, g in zip(ws, gs):
                        assert np.array_equal(w, g), (seed, "sse")
