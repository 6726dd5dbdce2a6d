#!/usr/bin/env python3
"""Finetune a speech recognition model on the MI355X engine.

Usage (same key=value surface as the reference's Hydra script, R/src/scripts/finetune_asr_model.py):
    python scripts/finetune_asr_model.py model=wav2vec2-small datasets=synthetic max_steps=10
    python -m torch.distributed.run --nproc-per-node 8 scripts/finetune_asr_model.py model=wav2vec2-large ...
"""
import logging
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch  # noqa: E402

from coral_amd.config import load_config  # noqa: E402
from coral_amd.finetune import finetune  # noqa: E402

logger = logging.getLogger("coral_amd")


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(levelname)s] <%(name)s> %(message)s")
    config = load_config("asr_finetuning", list(argv if argv is not None else sys.argv[1:]))
    is_main = os.getenv("RANK", "0") == "0"
    world = os.getenv("WORLD_SIZE")
    if world is not None:
        # multi-GPU rules of the reference (R/src/scripts/finetune_asr_model.py:46-61)
        if "layerdrop" in config.model and config.model.layerdrop != 0.0:
            if is_main:
                logger.info("Forcing `layerdrop` to 0.0 as this is required in a multi-GPU training")
            config.model.layerdrop = 0.0
        if config.model.type == "wav2vec2" and config.padding != "max_length":
            if is_main:
                logger.info("Forcing `padding` to 'max_length' as this is required in a multi-GPU training "
                            "with Wav2Vec 2.0 models")
            config.padding = "max_length"
        if int(world) > 1 and not torch.distributed.is_initialized():
            torch.cuda.set_device(int(os.getenv("LOCAL_RANK", "0")))
            torch.distributed.init_process_group("nccl")
    return finetune(config)


if __name__ == "__main__":
    main()
