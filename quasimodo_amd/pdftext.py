"""A PDF of plain text lines, written from the format description (PDF 1.4: catalog, page tree, one content stream per
page, the built-in Courier font) -- no plotting library exists in the build image and none is needed for this.

The reference's rules declare FIGURES among their outputs (rules/compare_FP.smk:10 `snpcaller_fp_snp_compare.pdf`,
eval_variant_custom.smk:83 `snpcall_benchmark.pdf`), drawn by R (VennDiagram / ggplot2).  Drawing is out of this path's
scope, but a Snakemake rule must leave every output it declares: the re-authored
rules (rules/*.smk of this repository) write the NUMBERS the figure would show as a table in a PDF under the declared name,
beside the machine-readable table.  Whoever wants the drawing runs the reference's R script on the same inputs."""


def _esc(s):
    return s.replace("\\", "\\\\").replace("(", "\\(").replace(")", "\\)")


MARKER = "[qmvt: a text-only stand-in for the figure -- the numbers of the rule's table; drawing is R's (not rebuilt)]"


def write_text_pdf(path, pages, title=""):
    """pages: list of lists of text lines (ASCII; anything else is replaced by '?').  Letter-size pages, Courier 9 pt,
    65 lines per page at most (longer pages are split) behind a marker line that says what the file is (nobody
    should take it for the reference's plot).  Deterministic bytes: no dates, no ids."""
    per_page = 65       # + the marker line every page starts with
    flat = []
    for lines in pages or [[]]:
        lines = [str(x) for x in lines] or [""]
        for i in range(0, len(lines), per_page):
            flat.append([MARKER] + lines[i:i + per_page])
    objs = []   # object k + 1 = objs[k] (bytes, without the "n 0 obj" frame)

    def add(body):
        objs.append(body)
        return len(objs)

    font = add(b"<< /Type /Font /Subtype /Type1 /BaseFont /Courier >>")
    pages_id = len(objs) + 1 + 2 * len(flat)      # the page tree comes after every page and its stream
    kids = []
    for lines in flat:
        ops = ["BT", "/F1 9 Tf", "11 TL", "40 750 Td"]
        for ln in lines:
            ops.append("(%s) Tj T*" % _esc(ln.encode("ascii", "replace").decode("ascii")))
        ops.append("ET")
        stream = "\n".join(ops).encode("ascii")
        sid = add(b"<< /Length %d >>\nstream\n" % len(stream) + stream + b"\nendstream")
        kids.append(add(b"<< /Type /Page /Parent %d 0 R /MediaBox [0 0 612 792] /Contents %d 0 R "
                        b"/Resources << /Font << /F1 %d 0 R >> >> >>" % (pages_id, sid, font)))
    got = add(b"<< /Type /Pages /Kids [%s] /Count %d >>" % (b" ".join(b"%d 0 R" % k for k in kids), len(kids)))
    assert got == pages_id
    catalog = add(b"<< /Type /Catalog /Pages %d 0 R >>" % pages_id)
    info = add(b"<< /Title (%s) /Producer (quasimodo_amd) >>" % _esc(title).encode("ascii", "replace"))
    out = bytearray(b"%PDF-1.4\n")
    offs = []
    for k, body in enumerate(objs):
        offs.append(len(out))
        out += b"%d 0 obj\n" % (k + 1) + body + b"\nendobj\n"
    xref = len(out)
    out += b"xref\n0 %d\n" % (len(objs) + 1) + b"0000000000 65535 f \n"
    for o in offs:
        out += b"%010d 00000 n \n" % o
    out += b"trailer\n<< /Size %d /Root %d 0 R /Info %d 0 R >>\nstartxref\n%d\n%%%%EOF\n" % (len(objs) + 1, catalog, info, xref)
    tmp = str(path) + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(bytes(out))
    import os
    os.replace(tmp, path)
    return len(flat)
