"""Host side of the ChemicalOCR stage: generated text -> OCR cells (words + boxes normalised to [0, 1]).

What the reference does between its two stages (markushgrapher/ocr/chemical_ocr.py:165-222: `parse_ocr_string`,
`clean_ocr_text`), restated as plain scanners so that the cells can go straight into `assembly` (word boxes, collation) without the
reference's intermediate dataset on disk (ref: inference.sh:165-184).  Pinned on outputs of the reference's own functions
(tools/make_golden_ocrtext.py -> tests/golden/host_ocrtext.json).  String work, microseconds per page; nothing here is on the GPU.

Grammar (chemical_ocr.py:165-199): optional <ocr> ... </ocr> wrapper; then either the legacy form, lines of
`<loc_x1><loc_y1><loc_x2><loc_y2>text` after an optional page box `<loc_0><loc_0><loc_500><loc_500>`, or the current form, lines of
`[page box>]x1>y1>x2>y2>text` with integer coordinates on a 500-unit grid.
"""
from __future__ import annotations

from typing import List, Tuple

_PAGE = "<loc_0><loc_0><loc_500><loc_500>"


def clean_ocr_text(text: str, start_tag: str = "<ocr>", end_tag: str = "</ocr>") -> str:
    """Drop everything before the first start tag and after the first end tag (both kept); a missing tag leaves that side alone.
    As the reference's `re.sub(r"(</ocr>).*?$", ...)` without MULTILINE, whose `$` also matches in front of a final newline: a
    text that ends with a newline keeps that one newline behind the end tag."""
    i = text.find(start_tag)
    if i >= 0:
        text = text[i:]
    if end_tag:
        j = text.find(end_tag)
        if j >= 0:
            text = text[:j + len(end_tag)] + ("\n" if text.endswith("\n") else "")
    return text


def _loc_numbers(line: str) -> List[int]:
    out, i = [], 0
    while True:
        i = line.find("<loc_", i)
        if i < 0:
            return out
        j = i + 5
        k = j
        while k < len(line) and line[k].isdecimal():
            k += 1
        if k > j and k < len(line) and line[k] == ">":
            out.append(int(line[j:k]))
            i = k + 1
        else:
            i = j


def _strip_loc_quads(line: str) -> str:
    """Remove every run of exactly four consecutive <loc_N> tags (left to right, non-overlapping)."""
    def tag_end(s, i):                      # index after a <loc_N> tag starting at i, or -1
        if not s.startswith("<loc_", i):
            return -1
        k = i + 5
        while k < len(s) and s[k].isdecimal():
            k += 1
        return k + 1 if k > i + 5 and k < len(s) and s[k] == ">" else -1
    out, i = [], 0
    while i < len(line):
        e, n, j = i, 0, i
        while n < 4:
            e = tag_end(line, j)
            if e < 0:
                break
            j, n = e, n + 1
        if n == 4:
            i = j
        else:
            out.append(line[i])
            i += 1
    return "".join(out)


def parse_ocr_string(ocr_string: str) -> Tuple[List[str], List[List[float]]]:
    cleaned = ocr_string.replace("</ocr>", "").replace("<ocr>", "").strip()
    words, boxes = [], []
    if "<loc_" in cleaned:
        if cleaned.startswith(_PAGE):
            cleaned = cleaned[len(_PAGE):].strip()
        for line in cleaned.splitlines():
            locs = _loc_numbers(line)
            text = _strip_loc_quads(line).strip()
            if len(locs) >= 4 and text:
                words.append(text)
                boxes.append([x / 500 for x in locs[-4:]])
        return words, boxes
    for line in cleaned.splitlines():
        line = line.strip()
        # leading integer fields "N>"; the LAST four of them that still leave a non-empty remainder are the box
        fields, i = [], 0
        while True:
            k = i
            while k < len(line) and line[k].isdecimal():
                k += 1
            if k > i and k < len(line) and line[k] == ">":
                fields.append((i, k + 1))
                i = k + 1
            else:
                break
        # the remainder must hold at least one character: give fields back until it does
        while fields and fields[-1][1] >= len(line):
            fields.pop()
        if len(fields) < 4:
            continue
        rest = line[fields[-1][1]:]
        if "\n" in rest or not rest:
            continue
        x1, y1, x2, y2 = (int(line[a:b - 1]) for a, b in fields[-4:])
        text = rest.strip()
        if text:
            words.append(text)
            boxes.append([x1 / 500, y1 / 500, x2 / 500, y2 / 500])
    return words, boxes
