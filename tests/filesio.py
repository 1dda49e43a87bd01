"""Writes a synth.small dataset as real files (sites VCF, DNM VCF / BED, PED, BAM) so the
file decoders and the command line can be exercised end to end.  Test infrastructure."""
import gzip
import os

from unfazed_amd.io_bam import write_bam

GT_TEXT = {0: "0/0", 1: "0/1", 2: "./.", 3: "1/1"}


def _num(x):
    if x is None or x < 0:
        return "."
    return ("%g" % x)


def vcf_text(samples, records, contigs):
    lines = ["##fileformat=VCFv4.2"]
    lines += ["##contig=<ID=%s>" % c for c in contigs]
    lines += ['##INFO=<ID=SVTYPE,Number=1,Type=String,Description="sv type">',
              '##INFO=<ID=END,Number=1,Type=Integer,Description="end">',
              '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
              '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="Allelic depths">',
              '##FORMAT=<ID=GQ,Number=1,Type=Float,Description="Genotype quality">']
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples))
    for r in records:
        info = ";".join("%s=%s" % (k, v) for k, v in r.info.items()) or "."
        cols = []
        for i in range(len(samples)):
            rd, ad = r.ref_depths[i], r.alt_depths[i]
            adtxt = "." if (rd < 0 and ad < 0) else ",".join([_num(rd), _num(ad)] + ["0"] * (len(r.alts) - 1))
            cols.append("%s:%s:%s" % (GT_TEXT[int(r.gt_types[i])], adtxt, _num(float(r.gt_quals[i]))))
        lines.append("\t".join([r.chrom, str(r.start + 1), ".", r.ref, ",".join(r.alts) if r.alts else ".", "50", "PASS",
                                info, "GT:AD:GQ"] + cols))
    return "\n".join(lines) + "\n"


def dump_dataset(ds, outdir, contig_len=10_000_000):
    os.makedirs(outdir, exist_ok=True)
    paths = {}
    paths["sites"] = os.path.join(outdir, "sites.vcf.gz")
    with gzip.open(paths["sites"], "wt") as fh:
        fh.write(vcf_text(ds.samples, ds.sites, ds.contigs))
    # DNM VCF: the DNMs' own records (first record at each DNM position)
    seen = set()
    dn_recs = []
    want = {(d["chrom"], d["start"]) for d in ds.dnms}
    for r in ds.sites:
        k = (r.chrom, r.start)
        if k in want and k not in seen:
            seen.add(k)
            dn_recs.append(r)
    paths["dnm_vcf"] = os.path.join(outdir, "dnms.vcf")
    with open(paths["dnm_vcf"], "w") as fh:
        fh.write(vcf_text(ds.samples, dn_recs, ds.contigs))
    paths["dnm_bed"] = os.path.join(outdir, "dnms.bed")
    with open(paths["dnm_bed"], "w") as fh:
        fh.write("#chrom\tstart\tend\tkid_id\tvar_type\n")
        for d in ds.dnms:
            fh.write("%s\t%d\t%d\t%s\t%s\n" % (d["chrom"], d["start"], d["end"], d["kid"], "SNV"))
    paths["ped"] = os.path.join(outdir, "trio.ped")
    with open(paths["ped"], "w") as fh:
        fh.write("#Family-ID\tIndividual-ID\tPaternal-ID\tMaternal-ID\tGender\n")
        for kid, p in ds.pedigrees.items():
            fh.write("F\t%s\t%s\t%s\t%s\n" % (kid, p["dad"], p["mom"], p["sex"]))
            fh.write("F\t%s\t0\t0\t1\nF\t%s\t0\t0\t2\n" % (p["dad"], p["mom"]))
    paths["bams"] = {}
    for kid, segs in ds.reads.items():
        paths["bams"][kid] = os.path.join(outdir, "%s.bam" % kid)
        write_bam(paths["bams"][kid], [(c, contig_len) for c in ds.contigs], segs)
    return paths
