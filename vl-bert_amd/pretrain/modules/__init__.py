from .resnet_vlbert_for_pretraining import ResNetVLBERTForPretraining, ResNetVLBERTForPretrainingMultitask  # noqa: F401
