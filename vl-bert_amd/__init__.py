"""MI355X-native VL-BERT pre-training hot path.

The directory is named `vl-bert_amd` (hyphenated), so import it with
`importlib.import_module("vl-bert_amd")` or via the root-level alias
`import vlbert_amd`.
"""
__version__ = "0.1.0"
