"""The product route for a LARGE batch of one kid from an indexed BAM (hostpath._chunked_batch): chunks of DNMs, chunk k + 1 decoded on
a worker thread while chunk k is uploaded and phased.  The records must be exactly those of the one-table route -- DNMs are independent
(reference: one task per DNM, snv_phaser.py:244-298) -- and those of the CPU oracle through the same host code."""
import contextlib
import io
import os

import pytest

from helpers import norm_records
from synth.small import SmallConfig, make_small

pytestmark = pytest.mark.gpu


def _run(paths, ds, env, readlen=151):
    from unfazed_amd import session
    from unfazed_amd.snv_phaser import phase_snvs
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    session._READS.clear()
    session._HOSTS.clear()
    for k in [k for k in session._SITES if "@" in k]:
        del session._SITES[k]
    try:
        kid = ds.dnms[0]["kid"]
        dnms = [dict(chrom=d["chrom"], start=d["start"], end=d["end"], kid=d["kid"], vartype="POINT", bam=paths["bams"][d["kid"]], cram_ref=None)
                for d in ds.dnms if d["kid"] == kid]
        err = io.StringIO()
        with contextlib.redirect_stderr(err):
            recs = phase_snvs(dnms, [kid], ds.pedigrees, paths["sites"], 2, "38", False, 10 ** 9, False, [0.0, 0.2], [0.8, 1.0], [0.2, 0.8], 20, 10, 5000,
                              1000000, 3, 1, readlen, 5)
        return norm_records(recs), sorted(err.getvalue().splitlines()), len(dnms)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_chunked_route_equals_one_table_route(tmp_path, hip_lib):
    import gzip
    from filesio import dump_dataset, write_bai, write_bgzf_text, write_tbi
    ds = make_small(SmallConfig(seed=909, n_dnms=46, cluster_prob=0.5))  # (one kid: the chunked route takes a batch of one alignment file)
    paths = dump_dataset(ds, str(tmp_path))
    for b in paths["bams"].values():
        write_bai(b)
    text = gzip.open(paths["sites"], "rt").read()
    write_bgzf_text(paths["sites"], text)
    write_tbi(paths["sites"])
    # the one-table route with the HOST's walk (the link form) is the yardstick; every route below walks the records on the device unless it says otherwise
    whole, err_w, n = _run(paths, ds, {"UZ_HOST_CHUNKS": "0", "UZ_WALK": "host"})
    assert n >= 40 and len(whole) >= 8
    dev, err_d, _ = _run(paths, ds, {"UZ_HOST_CHUNKS": "0", "UZ_WALK": "device"})  # one table, walked on the device (include/uz_bamwalk.h)
    assert dev == whole and err_d == err_w
    from unfazed_amd import hostpath
    calls = []
    orig = hostpath.PhasingHost._chunked_batch

    def spy(self, *a, **k):
        r = orig(self, *a, **k)
        calls.append(r)
        return r
    hostpath.PhasingHost._chunked_batch = spy
    try:
        for size, walk in (("5", "device"), ("13", "device"), ("13", "host")):
            got, err_g, _ = _run(paths, ds, {"UZ_HOST_CHUNKS": "1", "UZ_HOST_CHUNK_DNMS": size, "UZ_WALK": walk})
            assert got == whole, (size, walk)
            assert err_g == err_w, (size, walk)
    finally:
        hostpath.PhasingHost._chunked_batch = orig
    assert calls == [True, True, True]  # (the chunked route really ran)


def test_long_reads_travel_as_lists_with_two_byte_positions(tmp_path, hip_lib):
    """Reads longer than 256 bases: the listed bases of a record carry two-byte positions (bl_wide).  From files through the product route, with
    the list form and with every staged unit as a row (UZ_BASE_LISTS=0): the same records."""
    import gzip
    from filesio import dump_dataset, write_bai, write_bgzf_text, write_tbi
    from unfazed_amd import io_native
    ds = make_small(SmallConfig(seed=77, n_dnms=14, readlen=300, ins_mean=800.0, ins_sd=60.0, coverage_per_hap=18.0, cluster_prob=0.6))
    paths = dump_dataset(ds, str(tmp_path))
    for b in paths["bams"].values():
        write_bai(b)
    text = gzip.open(paths["sites"], "rt").read()
    write_bgzf_text(paths["sites"], text)
    write_tbi(paths["sites"])
    seen = []
    orig = io_native.BamSource.select

    def spy(self, *a, **k):
        r = orig(self, *a, **k)
        seen.append((int(r.view.n_bl), int(r.view.bl_wide), int(r.view.n_seq_units)))
        return r
    io_native.BamSource.select = spy
    try:
        rows, err_r, n = _run(paths, ds, {"UZ_BASE_LISTS": "0", "UZ_HOST_CHUNKS": "0", "UZ_WALK": "host"}, readlen=300)
        lists, err_l, _ = _run(paths, ds, {"UZ_BASE_LISTS": "1", "UZ_HOST_CHUNKS": "0", "UZ_WALK": "host"}, readlen=300)
        walked, err_w, _ = _run(paths, ds, {"UZ_HOST_CHUNKS": "0", "UZ_WALK": "device"}, readlen=300)  # (the device's walk: no link form at all)
    finally:
        io_native.BamSource.select = orig
    assert len(rows) >= 3 and lists == rows and err_l == err_r
    assert walked == rows and err_w == err_r and len(seen) == 2
    assert seen[0][0] == 0 and seen[1][0] > 100 and seen[1][1] == 1 and seen[1][2] < seen[0][2]


def test_two_threads_phase_at_once(tmp_path, hip_lib):
    """phase_snvs from two threads of one process (the call toggled the interpreter's collector per call and shared the device context):
    the calls take turns on the device (session.DEVICE_LOCK), the collector's pause is counted across them, both get the records a lone call gets,
    with the device's joins (the default) and with the host's"""
    import gc
    import gzip
    import threading
    from filesio import dump_dataset, write_bai, write_bgzf_text, write_tbi
    ds = make_small(SmallConfig(seed=911, n_dnms=30, cluster_prob=0.5))
    paths = dump_dataset(ds, str(tmp_path))
    for b in paths["bams"].values():
        write_bai(b)
    text = gzip.open(paths["sites"], "rt").read()
    write_bgzf_text(paths["sites"], text)
    write_tbi(paths["sites"])
    alone, err_a, n = _run(paths, ds, {"UZ_HOST_CHUNKS": "1", "UZ_HOST_CHUNK_DNMS": "7"})
    host_joins, err_h, _ = _run(paths, ds, {"UZ_HOST_CHUNKS": "1", "UZ_HOST_CHUNK_DNMS": "7", "UZ_JOINS": "host"})
    assert host_joins == alone and err_h == err_a and len(alone) >= 4
    from unfazed_amd.snv_phaser import phase_snvs
    kid = ds.dnms[0]["kid"]
    out, errs = [None, None], []

    def call(k):
        try:
            dnms = [dict(chrom=d["chrom"], start=d["start"], end=d["end"], kid=d["kid"], vartype="POINT", bam=paths["bams"][d["kid"]], cram_ref=None)
                    for d in ds.dnms if d["kid"] == kid]
            out[k] = norm_records(phase_snvs(dnms, [kid], ds.pedigrees, paths["sites"], 2, "38", False, 10 ** 9, True, [0.0, 0.2], [0.8, 1.0], [0.2, 0.8], 20, 10,
                                             5000, 1000000, 3, 1, 151, 5))
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    os.environ["UZ_HOST_CHUNKS"], os.environ["UZ_HOST_CHUNK_DNMS"] = "1", "7"
    try:
        assert gc.isenabled()
        ts = [threading.Thread(target=call, args=(k,)) for k in (0, 1)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        os.environ.pop("UZ_HOST_CHUNKS", None)
        os.environ.pop("UZ_HOST_CHUNK_DNMS", None)
    assert not errs, errs
    assert out[0] == alone and out[1] == alone
    assert gc.isenabled()
