"""`python -m vl-bert_amd.vqa.train_end2end --cfg cfgs/vqa/large_4x16G_fp32.yaml [--dist]` -- the reference's vqa/train_end2end.py:12-60
over the MI355X module mirror (vl-bert_amd/vqa/modules/resnet_vlbert_for_vqa.py); the loop is vl-bert_amd/common/finetune_entry.py."""
import sys

from ..common.finetune_entry import main as _main


def main(argv=None):
    return _main("vqa", argv)


if __name__ == "__main__":
    main(sys.argv[1:])
