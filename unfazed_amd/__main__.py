"""Command line: the reference's flags and defaults (reference unfazed/__main__.py:19-243)."""
from __future__ import annotations

import argparse
import os
import sys

from . import __reference_version__, __version__
from .unfazed import unfazed


def pair(arg):
    return [x for x in arg.split(":")]


def float_pair(arg):
    return [float(x) for x in arg.split(":")]


def setup_args():
    p = argparse.ArgumentParser(prog="unfazed", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("-v", "--version", action="version",
                   help="Installed version ({})".format(__reference_version__),
                   version="%(prog)s {} (unfazed_amd {}, MI355X)".format(__reference_version__, __version__))
    p.add_argument("-d", "--dnms", required=True,
                   help="valid VCF OR BED file of the DNMs of interest> If BED, must contain chrom, start, end, kid_id, var_type columns")
    p.add_argument("-s", "--sites", required=True,
                   help="sorted/bgzipped/indexed VCF/BCF file of SNVs to identify informative sites. Must contain each kid and both parents")
    p.add_argument("-p", "--ped", type=str, required=True, help="ped file including the kid and both parent IDs")
    p.add_argument("-b", "--bam-dir", type=str, required=False,
                   help="directory where bam/cram files (named {sample_id}.bam or {sample_id}.cram) are stored for offspring. If not included, --bam-pairs must be set")
    p.add_argument("--bam-pairs", type=pair, nargs="*", required=False,
                   help="space-delimited list of pairs in the format {sample_id}:{bam_path} where {sample_id} matches an offspring id from the dnm file. Can be used with --bam-dir arg, must be used in its absence")
    p.add_argument("-t", "--threads", type=int, default=2, help="number of threads to use")
    p.add_argument("-o", "--output-type", type=str, choices=["vcf", "bed"],
                   help="choose output type. If --dnms is not a VCF/BCF, output must be to BED format. Defaults to match --dnms input file")
    p.add_argument("--include-ambiguous", action="store_true", default=False, help="include ambiguous phasing results")
    p.add_argument("--verbose", action="store_true", default=False,
                   help="print verbose output including sites and reads used for phasing. Only applies to BED output")
    p.add_argument("--outfile", default="/dev/stdout", help="name for output file. Defaults to stdout")
    p.add_argument("-r", "--reference", required=False, help="reference fasta file (required for crams)")
    p.add_argument("-g", "--build", choices=["37", "38", "na"], required=True, type=str,
                   help="human genome build, used to determine sex chromosome pseudoautosomal regions. If `na` option is chosen, sex chromosomes will not be auto-phased. HG19/GRCh37 interchangeable")
    p.add_argument("--no-extended", action="store_true", default=False,
                   help="do not perform extended read-based phasing (default True)")
    p.add_argument("--multiread-proc-min", type=int, default=1000,
                   help="min number of variants required to perform multiple parallel reads of the sites file")
    p.add_argument("-q", "--quiet", action="store_true", help="no logging of variant processing data")
    p.add_argument("--min-gt-qual", type=int, default=20, help="min genotype and base quality for informative sites")
    p.add_argument("--min-depth", type=int, default=10, help="min coverage for informative sites")
    p.add_argument("--ab-homref", type=float_pair, default="0.0:0.2",
                   help="allele balance range for homozygous reference informative sites")
    p.add_argument("--ab-homalt", type=float_pair, default="0.8:1.0",
                   help="allele balance range for homozygous alternate informative sites")
    p.add_argument("--ab-het", type=float_pair, default="0.2:0.8",
                   help="allele balance range for heterozygous informative sites")
    p.add_argument("--evidence-min-ratio", type=int, default="10",
                   help="minimum ratio of evidence for a parent to provide an unambiguous call. Default 10:1")
    p.add_argument("--search-dist", type=int, default=5000,
                   help="maximum search distance from variant for informative sites (in bases)")
    p.add_argument("--insert-size-max-sample", type=int, default=1000000,
                   help="maximum number of read inserts to sample in order to estimate concordant read insert size")
    p.add_argument("--min-map-qual", type=int, default=1, help="minimum map quality for reads")
    p.add_argument("--stdevs", type=int, default=3,
                   help="number of standard deviations from the mean insert length to define a discordant read")
    p.add_argument("--readlen", type=int, default=151, help="expected length of input reads")
    p.add_argument("--split-error-margin", type=int, default=5,
                   help="margin of error for the location of split read clipping in bases")
    p.add_argument("--max-reads", type=int, default=100,
                   help="maximum number of reads to collect for phasing a single variant")
    p.add_argument("--gpus", type=int, default=1,
                   help="(unfazed_amd) GPUs of this node to phase on: the DNM list is cut into contiguous shards, one process per GPU, no collective "
                        "on the data path; rank 0 writes the output.  Under torchrun the launcher's WORLD_SIZE decides")
    p.add_argument("--sv-allele-balance-only", action="store_true", default=False,
                   help="(unfazed_amd) phase DEL/DUP by allele balance only; read-backed SV evidence is not built yet")
    return p


def main():
    print("\nUNFAZED v{}".format(__reference_version__), file=sys.stderr)
    parser = setup_args()
    args = parser.parse_args()
    print("Genome Build: {}\n".format(args.build), file=sys.stderr)
    if args.bam_dir is None and args.bam_pairs is None:
        print("\nMissing required argument: --bam-dir or --bam-pairs must be set\n", file=sys.stderr)
        sys.exit(parser.print_help())
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))  # before anything touches a GPU: the ranks are children of this process
    unfazed(args)


def spawn_ranks(n: int) -> int:
    """`python -m unfazed_amd --gpus N` without a launcher: the N ranks as child processes (same command line, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment); rank 0 writes the output.  All are ended when one fails."""
    import socket
    import subprocess
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-m", "unfazed_amd"] + sys.argv[1:], env=env))
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.1)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = max(rc, abs(r))
                for q in live:
                    q.terminate()
    return rc


if __name__ == "__main__":
    sys.exit(main() or 0)
