"""Make `from transformers.models.markushgrapher import ...` (ref: markushgrapher/core/common/begin.py:7-13) resolve
to this package, so the reference's `begin.py` / `eval.py` / `utils_evaluation.py` run unchanged on the HIP engine.

    import markushgrapher_amd.hf_shim  # before importing markushgrapher.*

Model and config come from this package (the accelerated path).  Tokenizer / processor / image processor are host-side
text and PIL work outside the path; they are mapped to the stock UDOP classes the fork derives from
(UdopTokenizer, UdopProcessor, LayoutLMv3ImageProcessor) when `transformers` is installed.
"""
import sys
import types


def install():
    from .modeling import MarkushgrapherConfig, MarkushgrapherForConditionalGeneration
    mod = types.ModuleType("transformers.models.markushgrapher")
    mod.MarkushgrapherConfig = MarkushgrapherConfig
    mod.MarkushgrapherForConditionalGeneration = MarkushgrapherForConditionalGeneration
    try:
        import transformers
        from transformers import LayoutLMv3ImageProcessor, UdopProcessor, UdopTokenizer
        mod.MarkushgrapherImageProcessor = LayoutLMv3ImageProcessor
        mod.MarkushgrapherProcessor = UdopProcessor
        mod.MarkushgrapherTokenizer = UdopTokenizer
        import transformers.models as tm
        setattr(tm, "markushgrapher", mod)
    except Exception:   # transformers absent: model + config are still importable under the alias
        pass
    sys.modules["transformers.models.markushgrapher"] = mod
    return mod


install()
