"""`python -m vl-bert_amd.vcr.train_end2end --cfg cfgs/vcr/large_q2a_4x16G_fp16.yaml [--dist]` -- the reference's vcr/train_end2end.py
over the MI355X module mirror (vl-bert_amd/vcr/modules/resnet_vlbert_for_vcr.py); the loop is vl-bert_amd/common/finetune_entry.py."""
import sys

from ..common.finetune_entry import main as _main


def main(argv=None):
    return _main("vcr", argv)


if __name__ == "__main__":
    main(sys.argv[1:])
