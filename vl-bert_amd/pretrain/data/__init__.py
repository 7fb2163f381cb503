from .collate import BatchCollator
from .datasets import DATASET_CATALOGS, ConceptualCaptionsDataset, GeneralCorpus, default_tokenizer
from .loader import DistributedSampler, MultiTaskDataLoader, make_dataloader, make_dataloaders
from .records import decode_detector_record, encode_detector_record
from .transforms import build_transforms
