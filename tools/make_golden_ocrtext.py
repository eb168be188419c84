"""Mints tests/golden/host_ocrtext.json: inputs and outputs of the REFERENCE's own `parse_ocr_string` / `clean_ocr_text`
(/root/reference/markushgrapher/ocr/chemical_ocr.py:165-222), imported unmodified in the build container.  Only data is written.
    python tools/make_golden_ocrtext.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from markushgrapher.ocr.chemical_ocr import parse_ocr_string, clean_ocr_text  # noqa: E402

CASES = [
    "<ocr>0>0>500>500>10>20>30>40>CH3\n5>6>7>8> OH </ocr>",
    "10>20>30>40>R1 = H, Me\n11>21>31>41>n = 1-3",
    "0>0>500>500>12>13>14>15>3>4\n1>2>3>4>5>6>7>8>x>y",
    "<ocr>\n\n250>250>260>260>Cl\n\n</ocr>",
    "1>2>3>text with only three\n1>2>3>4>\n1>2>3>4>   \n>1>2>3>4>lead",
    "7>8>9>10>a>b>c\n 7>8>9>10>  spaced  ",
    "<loc_0><loc_0><loc_500><loc_500>\n<loc_10><loc_20><loc_30><loc_40>CH3\n<loc_1><loc_2><loc_3><loc_4> N ",
    "<ocr><loc_0><loc_0><loc_500><loc_500><loc_5><loc_6><loc_7><loc_8>OH\n<loc_5><loc_6><loc_7>short</ocr>",
    "<loc_1><loc_2><loc_3><loc_4><loc_5><loc_6><loc_7><loc_8>two quads\n<loc_9><loc_9><loc_9><loc_9>",
    "x<loc_1><loc_2>y<loc_3><loc_4><loc_5><loc_6>z",
    "",
    "no coordinates at all",
    "12>34>56>78>٣>arabic digit stays text",
    "0>0>500>500>",
    "500>500>500>500>X\r\n1>1>2>2>Y",
]
CLEAN = ["junk<ocr>abc</ocr>tail", "no tags", "<ocr>only start", "only end</ocr>x", "a<ocr>b<ocr>c</ocr>d</ocr>e", "x\n<ocr>multi\nline</ocr>\ny",
         "<ocr>a</ocr>\n", "<ocr>a</ocr>tail\n", "<ocr>a</ocr>t\n\n", "pre<ocr>a\n", "<ocr>a</ocr>\nmore", "\n<ocr>a</ocr>x\ny\n"]

out = {"parse": [], "clean": []}
for c in CASES:
    w, b = parse_ocr_string(c)
    out["parse"].append({"in": c, "words": w, "boxes": b})
for c in CLEAN:
    out["clean"].append({"in": c, "out": clean_ocr_text(c), "out_no_end": clean_ocr_text(c, end_tag=None)})
path = os.path.join(ROOT, "tests", "golden", "host_ocrtext.json")
with open(path, "w") as f:
    json.dump(out, f, ensure_ascii=False, indent=1)
print("wrote", path, len(out["parse"]), "parse cases,", len(out["clean"]), "clean cases")
