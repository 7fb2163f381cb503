from .resnet_vlbert_for_pretraining import ResNetVLBERTForPretraining  # noqa: F401
