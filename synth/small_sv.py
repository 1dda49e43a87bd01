"""Small-scale synthetic trio with de novo SVs (DEL / DUP / "INV"-typed) for the SV
read-backed path (collect_reads_sv) and the allele-balance path: split reads with SA
tags and supplementary pieces, discordant pairs spanning the event, soft-clipped reads at
the breakpoints, reads clipped at both ends (banned by the reference), plus the usual
het / informative sites around both breakpoints and inside the event.
Test and golden-vector infrastructure (record level, Python/numpy)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

from synth.small import BASES, SmallDataset, otherbase, refbase
from unfazed_amd.model import (FMREVERSE, FPAIRED, FPROPER, FREAD1, FREAD2, FREVERSE, FSUPP, OP_M, OP_S, Segment,
                               SiteRecord)


@dataclass
class SvConfig:
    seed: int = 1
    n_svs: int = 8
    contigs: List[str] = field(default_factory=lambda: ["1", "2"])
    spacing: int = 60000
    search_dist: int = 5000
    site_rate: float = 1.0 / 400.0
    coverage_per_hap: float = 15.0
    readlen: int = 151
    base_err: float = 0.004
    kid: str = "kid1"
    pad: int = 1200


def make_small_sv(cfg: SvConfig) -> SmallDataset:
    rng = np.random.RandomState(cfg.seed)
    kid, dad, mom = cfg.kid, "dad1", "mom1"
    samples = [mom, kid, dad]  # not in kid, dad, mom order
    col = {s: i for i, s in enumerate(samples)}
    ped = {kid: {"kid": kid, "dad": dad, "mom": mom, "sex": "2"}}
    L = cfg.readlen
    W = cfg.search_dist + cfg.pad
    per = (cfg.n_svs + len(cfg.contigs) - 1) // len(cfg.contigs)
    events = []
    for i in range(cfg.n_svs):
        tid = i // per
        s = 3 * W + (i % per) * cfg.spacing + int(rng.randint(0, 3000))
        length = int(rng.choice([900, 2500, 4000, 7000, 12000]))
        vt = ["DEL", "DUP", "DEL", "INV", "DUP"][i % 5]
        events.append(dict(tid=tid, start=s, end=s + length, vartype=vt, origin=int(rng.randint(0, 2)),
                           td=int(rng.randint(0, 2)), tm=int(rng.randint(0, 2))))
    sites: List[SiteRecord] = []
    ev_sites: List[Dict[int, tuple]] = []
    for ev in events:
        lo, hi = ev["start"] - W - 600, ev["end"] + W + 600
        n = rng.poisson((hi - lo) * cfg.site_rate)
        by_pos = {}
        for p in sorted(set(int(x) for x in rng.randint(lo, hi, size=n))):
            ref = refbase(ev["tid"], p)
            alt = otherbase(ref, p)
            f = float(rng.rand()) ** 2 * 0.9 + 0.05
            dh = [int(rng.rand() < f), int(rng.rand() < f)]
            mh = [int(rng.rand() < f), int(rng.rand() < f)]
            pat, mat = dh[ev["td"]], mh[ev["tm"]]
            gts, rds, ads, gqs = [0] * 3, [0] * 3, [0] * 3, [99.0] * 3
            inside = ev["start"] <= p < ev["end"]
            for smp, g in ((dad, dh[0] + dh[1]), (mom, mh[0] + mh[1])):
                depth = int(rng.poisson(30))
                a = int(rng.binomial(depth, (0.01, 0.5, 0.99)[g])) if depth else 0
                gts[col[smp]], rds[col[smp]], ads[col[smp]] = (0, 1, 3)[g], depth - a, a
            kg = pat + mat
            depth = int(rng.poisson(30))
            pa = (0.01, 0.5, 0.99)[kg]
            if inside and ev["vartype"] == "DEL":  # hemizygous: only the non-deleted haplotype is seen
                keep = mat if ev["origin"] == 0 else pat
                kg, pa, depth = 2 * keep, (0.01, 0.99)[keep], int(rng.poisson(15))
            elif inside and ev["vartype"] == "DUP" and kg == 1:  # 2:1 towards the duplicated haplotype
                dup_allele = pat if ev["origin"] == 0 else mat
                pa = 2 / 3 if dup_allele else 1 / 3
                depth = int(rng.poisson(45))
            a = int(rng.binomial(depth, pa)) if depth else 0
            gts[col[kid]], rds[col[kid]], ads[col[kid]] = (0, 1, 3)[kg], depth - a, a
            if rng.rand() < 0.05:
                gqs[int(rng.randint(0, 3))] = float(rng.randint(0, 40))
            sites.append(SiteRecord(cfg.contigs[ev["tid"]], p, ref, [alt], gts, rds, ads, gqs))
            by_pos[p] = (ref, alt, pat, mat)
        ev_sites.append(by_pos)
    sites.sort(key=lambda r: (cfg.contigs.index(r.chrom), r.start))

    reads: List[Segment] = []
    qn = 0

    BASE_ARR = np.frombuffer(BASES.encode(), np.uint8)

    def ref_codes(tid, a, n):  # refbase(tid, q) for q in [a, a + n), as 0..3 (the same integer hash, vectorised)
        q = np.arange(a, a + n, dtype=np.int64)
        x = (q * 0x9E3779B1 + tid * 0x85EBCA6B + 0x1234567) & 0xFFFFFFFF
        x ^= x >> 15
        x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
        x ^= x >> 12
        return (x & 3).astype(np.int64)

    site_keys = [np.array(sorted(d.keys()), np.int64) for d in ev_sites]

    bitgen = rng._bit_generator  # the MT19937 behind the legacy RandomState: rand() and randint() are functions of its raw words

    def bases(ev_i, tid, a, n, h):
        """n bases from a: the haplotype's allele at the event's sites, the reference elsewhere, with base errors.  The random
        draws are those of the plain loop -- one rand() per base, a randint(0, 3) right after a draw below base_err -- read off
        the generator's raw 32-bit words in blocks (rand() = two words, randint(0, 3) = words & 3 until one is not 3), so a
        read costs a few numpy calls instead of 151 Python-level draws and the stream stays word for word the same."""
        code = ref_codes(tid, a, n)
        ks = site_keys[ev_i]
        for q in ks[np.searchsorted(ks, a): np.searchsorted(ks, a + n)]:
            st = ev_sites[ev_i][int(q)]
            code[int(q) - a] = BASES.index(st[1] if st[2 + h] else st[0])
        raw = bitgen.random_raw(2 * n)
        at, i = 0, 0  # next unread word, next base
        while i < n:
            m = n - i
            short = 2 * m - (raw.size - at)
            if short > 0:
                raw = np.concatenate([raw[at:], bitgen.random_raw(short)])
                at = 0
            blk = raw[at: at + 2 * m]
            d = ((blk[0::2] >> np.uint64(5)) * np.uint64(67108864) + (blk[1::2] >> np.uint64(6))).astype(np.float64) / 9007199254740992.0
            hit = np.nonzero(d < cfg.base_err)[0]
            if hit.size == 0:
                at += 2 * m
                break
            j = int(hit[0])
            at += 2 * (j + 1)
            while True:  # randint(0, 3)
                if at >= raw.size:
                    raw = bitgen.random_raw(1)
                    at = 0
                v = int(raw[at]) & 3
                at += 1
                if v <= 2:
                    break
            code[i + j] = (code[i + j] + 1 + (v % 3)) % 4  # otherbase
            i += j + 1
        assert at == raw.size  # every word drawn was used: the generator stands where the plain loop would leave it
        return BASE_ARR[code].tobytes().decode()

    def rnd(n):
        return "".join(BASES[int(x)] for x in rng.randint(0, 4, size=n))

    def qual(n):
        return np.where(rng.rand(n) < 0.03, 12, 37).tolist()

    for ei, ev in enumerate(events):
        tid, s, e = ev["tid"], ev["start"], ev["end"]
        vt = ev["vartype"]
        lo, hi = s - W, e + W
        n_pairs = int(round(cfg.coverage_per_hap * (hi - lo) / (2.0 * L)))
        for h in (0, 1):
            carrier = h == ev["origin"]
            for _ in range(n_pairs):
                ins = int(np.clip(rng.normal(450, 50), 2 * L, 900))
                fs = int(rng.randint(lo - ins, hi))
                name = "s%07d" % qn
                qn += 1
                r1, r2 = fs, fs + ins - L  # reference starts of the two reads on an unaltered haplotype
                segs = []
                flag1 = FPAIRED | FPROPER | FREAD1 | FMREVERSE
                flag2 = FPAIRED | FPROPER | FREAD2 | FREVERSE
                jl = s if vt != "DUP" else e  # reference coordinate where the junction's left part ends
                jr = e if vt != "DUP" else s  # ... and where its right part starts
                made = False
                if carrier:
                    # place the fragment on the rearranged haplotype: coordinates left of the junction map
                    # 1:1, those right of it jump by (jr - jl)
                    x = int(rng.randint(-ins, ins)) if rng.rand() < 0.35 else None
                    if x is not None:
                        fa = jl + x - ins // 2  # fragment start in junction coordinates (junction at jl)
                        made = True
                        recs = []
                        for which, a in ((0, fa), (1, fa + ins - L)):
                            flag = (flag1 if which == 0 else flag2) & ~FPROPER
                            if a + L <= jl:  # wholly left of the junction
                                recs.append([which, flag, a, [(OP_M, L)], bases(ei, tid, a, L, h), False, None])
                            elif a >= jl:  # wholly right: shifted
                                ra = a - jl + jr
                                recs.append([which, flag, ra, [(OP_M, L)], bases(ei, tid, ra, L, h), False, None])
                            else:  # crosses the junction: split or clipped
                                k = jl - a
                                left_seq, right_seq = bases(ei, tid, a, k, h), bases(ei, tid, jr, L - k, h)
                                if k >= 25 and L - k >= 25:
                                    recs.append([which, flag, a, [(OP_M, k), (OP_S, L - k)], left_seq + right_seq, True, None])
                                    recs.append([which, flag | FSUPP, jr, [(OP_S, k), (OP_M, L - k)], left_seq + right_seq, True, None])
                                elif k >= L - k:
                                    recs.append([which, flag, a, [(OP_M, k), (OP_S, L - k)], left_seq + rnd(L - k), False, None])
                                else:
                                    recs.append([which, flag, jr, [(OP_S, k), (OP_M, L - k)], rnd(k) + right_seq, False, None])
                        prim = [r for r in recs if not (r[1] & FSUPP)]
                        p0, p1 = prim[0], prim[1]
                        tl = (p1[2] + sum(l for op, l in p1[3] if op == OP_M)) - p0[2]
                        for r in recs:
                            which = r[0]
                            other = p1 if which == 0 else p0
                            segs.append(Segment(name, r[1], tid, r[2], 60, r[3], tid, other[2], tl if which == 0 else -tl,
                                                r[4], qual(L), has_sa=r[5]))
                if not made:
                    u = rng.rand()
                    cig1 = [(OP_M, L)]
                    st1 = r1
                    if u < 0.02:  # clipped at both ends: start_matches < 7 and end_matches < 7 -> banned
                        cig1 = [(OP_S, 6), (OP_M, L - 12), (OP_S, 6)]
                        st1 = r1 + 6
                    seq1 = rnd(6) + bases(ei, tid, st1, L - 12, h) + rnd(6) if u < 0.02 else bases(ei, tid, r1, L, h)
                    segs.append(Segment(name, flag1, tid, st1, 0 if rng.rand() < 0.03 else 60, cig1, tid, r2, ins, seq1, qual(L)))
                    segs.append(Segment(name, flag2, tid, r2, 0 if rng.rand() < 0.03 else 60, [(OP_M, L)], tid, st1, -ins,
                                        bases(ei, tid, r2, L, h), qual(L)))
                reads.extend(segs)
    reads.sort(key=lambda sg: (sg.tid, sg.pos))
    dnms = [{"chrom": cfg.contigs[ev["tid"]], "start": ev["start"], "end": ev["end"], "kid": kid,
             "vartype": ev["vartype"], "bam": "mem://%s.bam" % kid, "cram_ref": None} for ev in events]
    truth = {"{chrom}_{start}_{end}_{kid}_{vartype}".format(**d): ("dad" if ev["origin"] == 0 else "mom")
             for d, ev in zip(dnms, events)}
    return SmallDataset(samples, list(cfg.contigs), sites, {kid: reads}, dnms, ped, truth)
