"""Mint tests/golden/frappe_compact.csv.gz: the Frappe app-usage log of the reference's data sets
(/root/reference/context-aware_data_sets/Mobile_Frappe.zip!Mobile_Frappe/frappe/frappe.csv, tab-separated) rewritten as the
comma-separated COMPACT rating format CARSKit reads (header = user,item,cnt,<8 context dimensions>; one cell per dimension) --
`CARSKit.validateDataFormat` and `DataTransformer` split on ',' only (CARSKit.java:188-189, DataTransformer.java:233-246), so the
TSV cannot be fed as it is.  Nothing else changes: every field, the line order and the raw usage count `cnt` as the rating.
This is DATA (BASELINE.json configs[1]: "CAMF_C k=64 fp32 on Frappe"), committed so the GPU box, which has no /root/reference,
can run the config on the real file.  License of the data set: tests/golden/frappe_LICENSE.txt (research use, cite
Baltrunas et al. 2015, arXiv:1505.03014; do not redistribute outside this test tree).

Run in the build container:  python tests/golden/make_frappe.py"""
import gzip
import io
import os
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
ZIP = "/root/reference/context-aware_data_sets/Mobile_Frappe.zip"


def main():
    z = zipfile.ZipFile(ZIP)
    raw = z.read("Mobile_Frappe/frappe/frappe.csv").decode("utf-8")
    assert "," not in raw and "\r" not in raw
    text = raw.replace("\t", ",")
    assert text.count("\n") == 96204 and text.split("\n", 1)[0] == "user,item,cnt,daytime,weekday,isweekend,homework,cost,weather,country,city"
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode="wb", compresslevel=9, mtime=0) as g:   # mtime=0: byte-reproducible
        g.write(text.encode("utf-8"))
    open(os.path.join(HERE, "frappe_compact.csv.gz"), "wb").write(buf.getvalue())
    open(os.path.join(HERE, "frappe_LICENSE.txt"), "w").write(z.read("Mobile_Frappe/README.txt").decode("utf-8"))
    print("wrote frappe_compact.csv.gz", len(buf.getvalue()), "bytes")


if __name__ == "__main__":
    main()
