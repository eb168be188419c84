"""Mints tests/golden/host_cxsmiles_opt.json: inputs and outputs of the REFERENCE's own inline lines that turn the decoded text into the
CXSMILES "opt" string (/root/reference/markushgrapher/utils/ocsr/utils_evaluation.py:306-352, inside get_smiles_metrics), executed
unmodified in the build container.

The enclosing function cannot be imported (rdkit / markushgenerator / cv2 are absent), so the slice of lines is READ from the
reference file at run time, dedented and exec'd in a harness that supplies the names it uses: `config`, `predicted_text`, `verbose`,
`re`, and a `cxsmiles_tokenizer_training` whose convert_opt_to_out (RDKit; stays in the reference) records its argument.  Nothing of
the reference's text is stored: only input / output pairs are written.
    python tools/make_golden_cxsmiles_opt.py
"""
import json
import os
import re
import textwrap
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/markushgrapher/utils/ocsr/utils_evaluation.py"
FIRST, LAST = 306, 352          # `if config["name"] == "ocsr":` ... `predicted_smiles_opt = None`

CASES = [
    "<markush><cxsmi> C C </cxsmi> <stable>x</stable>",
    "<cxsmi>C1=CC=CC=C1 |m:1:2.3|</cxsmi></s>",
    "<cxsmi> [R1] c 1 c c c c c 1 </cxsmi><r>R1 : Me , Et</r></s>",
    "no tags at all",
    "<cxsmi>first</cxsmi> text <cxsmi>second</cxsmi>",
    "<cxsmi></cxsmi>",
    "<cxsmi>unterminated C C",
    "C C </cxsmi> only the closing tag",
    "<smi>C O</smi></s>",
    "<smi> c 1 c c c c c 1 </smi> </s> </s>",
    "<cxsmi>C(=O)O</s></cxsmi>",
    "  <cxsmi>  N  </cxsmi>  ",
    "<cxsmi>a<cxsmi>b</cxsmi>c</cxsmi>",
    "<markush><cxsmi>*C* |$R1;;R2$|</cxsmi><stable><n>R1<ns>alkyl</stable></markush></s>",
    "",
]


def ref_slice():
    lines = open(REF).read().split("\n")[FIRST - 1:LAST]
    src = textwrap.dedent("\n".join(lines))
    assert src.startswith('if config["name"] == "ocsr":') and src.rstrip().endswith("predicted_smiles_opt = None"), "the reference lines moved"
    return compile(src, REF + f":{FIRST}-{LAST}", "exec")


def run(code, task, text):
    seen = []
    ct = types.SimpleNamespace(convert_opt_to_out=lambda s: (seen.append(s), s)[1])
    env = {"config": {"name": task}, "predicted_text": text, "verbose": False, "re": re, "cxsmiles_tokenizer_training": ct, "print": lambda *a, **k: None}
    exec(code, env)
    if task == "ocsr":
        return env.get("predicted_smiles")
    return env.get("predicted_smiles_opt")


if __name__ == "__main__":
    code = ref_slice()
    out = []
    for task in ("ocsr", "ocxsr", "mdu"):
        for text in CASES:
            out.append({"task": task, "text": text, "opt": run(code, task, text)})
    path = os.path.join(ROOT, "tests", "golden", "host_cxsmiles_opt.json")
    with open(path, "w") as f:
        json.dump({"source": f"utils_evaluation.py:{FIRST}-{LAST} exec'd unmodified", "cases": out}, f, indent=1)
    print("wrote", path, len(out), "cases;", sum(c["opt"] is None for c in out), "with no opt string")
