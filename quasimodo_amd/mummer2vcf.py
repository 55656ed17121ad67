"""Truth-set builder: MUMmer `show-snps -CTHlr` table -> truth VCF.

Restates the behaviour of the reference's program/mummer2vcf.py (run by rules/genome_diff.smk:22-24
as `mummer2vcf.py -s <table> --output-header -n -g <ref.fa>`) without Biopython:

  * one VCF row per table row: CHROM = ref tag (col 11), POS = P1 (col 1), REF/ALT = the two SUB
    columns, QUAL 30, FILTER PASS, INFO DP=30;REF1=..;REF2=..   (mummer2vcf.py:69-85)
  * -n: rows with an N/n in REF or ALT are dropped                (:88-100)
  * single-base, dot-free REF and ALT -> SNV, anything else INDEL  (:103-118)
  * SNVs are ordered by position (stable); rows at the position of the previous row fold their
    ALT into the kept row's comma list (no duplicates)             (:122-144, :242-252)
  * indels keep input order; a row continues the previous one when it is an insertion at the
    same position or a deletion at the next position: inserted bases are appended to every
    allele (a different query position) or add an alternative allele whose last base is
    replaced (the same query position); deleted bases are appended to REF   (:147-185)
  * every indel gets the reference base before it as anchor and its POS moves one to the left
    (VCF convention)                                               (:188-210)
  * rows are ordered by (CHROM, POS), SNVs before indels at equal keys, INFO gains
    ;ORIG=<query tag>:<P2>;TYPE=SNV|INDEL, eight columns are written  (:277-313)
  * --output-header: VCFv4.2 header with the contigs that carry variants, in FASTA order  (:320-357)

The reference cannot be executed in the build image (Bio is not installed), so parity is pinned
by hand-derived cases only (tests/test_mummer2vcf.py).

`convert` / the command line go through the library (`qm_mummer2vcf`, csrc/qmvt_host.cpp: SURVEY.md 8f-2 asks for the
builder in C++); `convert_py` below is the same restatement in Python, kept as the second opinion the tests compare the
library with on random tables."""
import sys
from time import strftime


def read_fasta(path):
    """{first word of the header: sequence}, in file order."""
    seqs, name, parts = {}, None, []
    with open(path) as fh:
        for line in fh:
            if line.startswith(">"):
                if name is not None:
                    seqs[name] = "".join(parts)
                name, parts = line[1:].split()[0] if line[1:].split() else "", []
            else:
                parts.append(line.strip())
    if name is not None:
        seqs[name] = "".join(parts)
    return seqs


class _Row:
    __slots__ = ("chrom", "pos", "ref", "alt", "info", "kind", "orig")

    def __init__(self, chrom, pos, ref, alt, info, kind, orig):
        self.chrom, self.pos, self.ref, self.alt, self.info, self.kind, self.orig = chrom, pos, ref, alt, info, kind, orig


def _parse(lines, no_ns):
    rows = []
    for line in lines:
        c = line.rstrip("\n\r\b").split("\t")
        ref, alt = c[1], c[2]
        if no_ns and ("N" in ref or "n" in ref or "N" in alt or "n" in alt):
            continue
        snv = len(ref) == 1 and "." not in ref and len(alt) == 1 and "." not in alt
        rows.append(_Row(c[10], int(c[0]), ref, alt, "DP=30;REF1=%s;REF2=%s" % (c[10], c[11]), "SNV" if snv else "INDEL",
                         "%s:%s" % (c[11], c[3])))
    return rows


def _fold_snvs(rows):
    out, prev_pos = [], None
    for r in sorted(rows, key=lambda r: r.pos):          # stable
        if out and r.pos == prev_pos:
            alts = out[-1].alt.split(",")
            if r.alt not in alts:
                out[-1].alt = ",".join(alts + [r.alt])
        else:
            out.append(r)
        prev_pos = r.pos
    return out


def _merge_indels(rows):
    out, prev_pos, prev_orig = [], None, None
    for r in rows:
        cont = bool(out) and ((r.pos == prev_pos and r.ref == ".") or (r.pos == prev_pos + 1 and r.alt == "."))
        if cont:
            last = out[-1]
            if r.ref == ".":
                if r.orig != prev_orig:
                    last.alt = ",".join(a + r.alt for a in last.alt.split(","))
                else:
                    last.alt = last.alt + "," + last.alt[:-1] + r.alt
            elif r.alt == ".":
                last.ref = last.ref + r.ref
        else:
            out.append(r)
        prev_pos, prev_orig = r.pos, r.orig
    return out


def _anchor(rows, seqs):
    for r in rows:
        base = seqs[r.chrom][r.pos - 2]     # the base before the variant (python index: -1 wraps like the reference)
        if r.ref == ".":
            r.ref, r.alt = base, ",".join(base + a for a in r.alt.split(","))
        elif r.alt == ".":
            r.ref, r.alt = base + r.ref, base
        r.pos -= 1
    return rows


def convert(table_lines, reference=None, no_ns=False, vtype="ALL", output_header=False, input_header=False, file_date=None):
    """Returns the VCF as a list of lines (no newline).  table_lines: the table's lines (str or bytes, with their newlines) or its
    whole text; reference: path of the FASTA.  The work is the library's (qm_mummer2vcf)."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    if isinstance(table_lines, (bytes, str)):
        text = table_lines
    else:
        table_lines = list(table_lines)
        text = (b"" if table_lines and isinstance(table_lines[0], bytes) else "").join(table_lines)
    if isinstance(text, str):
        text = text.encode("utf-8", "surrogateescape")
    fasta = b""
    if reference:
        with open(reference, "rb") as fh:
            fasta = fh.read()
    flags = (1 if no_ns else 0) | (2 if output_header else 0) | (4 if input_header else 0) | {"ALL": 0, "SNP": 8, "INDEL": 16}[vtype]
    out, n = C.c_void_p(), C.c_size_t()
    rc = L.qm_mummer2vcf(text, len(text), fasta, len(fasta), None if reference is None else str(reference).encode(), flags,
                         None if file_date is None else file_date.encode(), C.byref(out), C.byref(n))
    if rc != 0:
        raise _lib.QmvtError(rc, "qm_mummer2vcf: a row with fewer than 12 columns, a position that is no integer, or an indel outside the reference sequences")
    try:
        data = C.string_at(out, n.value)
    finally:
        L.qm_free(out)
    return data.decode("utf-8", "surrogateescape").split("\n")[:-1] if data else []


def convert_py(table_lines, reference=None, no_ns=False, vtype="ALL", output_header=False, input_header=False):
    """The same in Python (the restatement the library's is checked against).  Returns the VCF as a list of lines."""
    lines = list(table_lines)
    if input_header:
        lines = lines[4:]
    rows = _parse([ln for ln in lines if ln.strip("\n\r")], no_ns)
    seqs = read_fasta(reference) if reference else {}
    snvs = _fold_snvs([r for r in rows if r.kind == "SNV"])
    indels = _anchor(_merge_indels([r for r in rows if r.kind == "INDEL"]), seqs) if any(r.kind == "INDEL" for r in rows) else []
    # the reference first orders by POS as text, then by (CHROM, int POS); both sorts are stable
    merged = sorted(snvs + indels, key=lambda r: str(r.pos))
    if vtype == "SNP":
        merged = [r for r in merged if r.kind == "SNV"]
    elif vtype == "INDEL":
        merged = [r for r in merged if r.kind == "INDEL"]
    merged.sort(key=lambda r: (r.chrom, r.pos))
    body = ["\t".join([r.chrom, str(r.pos), ".", r.ref, r.alt, "30", "PASS", "%s;ORIG=%s;TYPE=%s" % (r.info, r.orig, r.kind)])
            for r in merged]
    if not output_header:
        return body
    head = ["##fileformat=VCFv4.2", "##fileDate=%s" % strftime("%Y%m%d"), "##source=mummer2vcf.py", "##reference=%s" % reference]
    used = {r.chrom for r in merged}
    head += ["##contig=<ID=%s,length=%d>" % (name, len(seq)) for name, seq in seqs.items() if name in used]
    head += ['##INFO=<ID=DP,Number=1,Type=Integer,Description="Total depth of quality bases">',
             '##INFO=<ID=REF1,Number=1,Type=String,Description="The name of the 1st reference sequence">',
             '##INFO=<ID=REF2,Number=1,Type=String,Description="The name of the 2nd reference sequence">',
             '##INFO=<ID=ORIG,Number=1,Type=String,Description="The original position of variant at 2nd reference sequence">',
             '##INFO=<ID=TYPE,Number=1,Type=String,Description="Indicates that the variant is an INDEL or SNV.">',
             "\t".join(["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO"])]
    return head + body


def main(argv=None):
    import argparse
    p = argparse.ArgumentParser(description="Convert MUMmer `show-snps -T` output to VCF")
    p.add_argument("-s", "--snps", required=True)
    p.add_argument("--input-header", action="store_true")
    p.add_argument("-n", "--no-Ns", action="store_true")
    p.add_argument("-t", "--type", choices=["SNP", "INDEL", "ALL"], default="ALL")
    p.add_argument("--output-header", action="store_true")
    p.add_argument("-g", "--reference")
    a = p.parse_args(argv)
    if a.output_header and not a.reference:
        sys.exit("ERROR: --add-vcf-header requires --reference as well\n\n")
    with open(a.snps, "rb") as fh:
        out = convert(fh.read(), reference=a.reference, no_ns=a.no_Ns, vtype=a.type, output_header=a.output_header,
                      input_header=a.input_header)
    for line in out:
        sys.stdout.write(line + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
