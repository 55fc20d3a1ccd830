"""CLI counterpart of the reference's run.py (run.py:38-158): same flag names and defaults for everything the
model/trainer consume, plus the extensions of this build (--dtype, --num_classes, --synthetic sizes, data parallel via
torch.distributed.run).  Data is synthetic (the HF hub files and the MVSA/HFM datasets are not available offline):

    python -m d2r_amd.run --num_epochs 2 --batch_size 32 --train_samples 256
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m d2r_amd.run --batch_size 256
"""
from __future__ import annotations

import argparse
import logging
import os
import random

import numpy as np
import torch

logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s -   %(message)s", datefmt="%m/%d/%Y %H:%M:%S",
                    level=logging.INFO)
logger = logging.getLogger("d2r_amd.run")


def set_seed(seed=2023):
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def build_parser():
    p = argparse.ArgumentParser()
    # --- flags of the reference (run.py:39-84); unused ones are accepted and ignored like there
    p.add_argument("--bert_name", default="bert-base-uncased", type=str)
    p.add_argument("--vit_name", default="clip-vit-base-patch32", type=str)
    p.add_argument("--num_epochs", default=30, type=int)
    p.add_argument("--device", default="cuda", type=str)
    p.add_argument("--batch_size", default=32, type=int, help="GLOBAL batch (split over ranks under torch.distributed.run)")
    p.add_argument("--lr", default=3e-5, type=float)
    p.add_argument("--warmup_ratio", default=0.01, type=float)
    p.add_argument("--eval_begin_epoch", default=1, type=int)
    p.add_argument("--seed", default=2023, type=int)
    p.add_argument("--load_path", default=None, type=str)
    p.add_argument("--save_path", default="./output/", type=str)
    p.add_argument("--write_path", default=None, type=str)
    p.add_argument("--notes", default="", type=str)
    p.add_argument("--do_train", action="store_true", default=True)
    p.add_argument("--only_test", action="store_true")
    p.add_argument("--max_seq", default=128, type=int)
    p.add_argument("--ignore_idx", default=0, type=int)
    p.add_argument("--sample_ratio", default=1.0, type=float)
    p.add_argument("--alpha", default=0, type=float)
    p.add_argument("--margin", default=0.1, type=float)
    p.add_argument("--beta", default=0.1, type=float)
    p.add_argument("--mild_margin", default=0.7, type=float)
    p.add_argument("--hetero", default=0.9, type=float)
    p.add_argument("--homo", default=0.9, type=float)
    p.add_argument("--DR_step", default=3, type=int)
    p.add_argument("--weight_js_1", default=0.1, type=float)
    p.add_argument("--weight_js_2", default=0.1, type=float)
    p.add_argument("--weight_diff", default=0.1, type=float)
    p.add_argument("--embed_size", default=768, type=int)
    p.add_argument("--num_head_IMRC", type=int, default=16)
    p.add_argument("--hid_IMRC", type=int, default=768)
    p.add_argument("--raw_feature_norm_CMRC", default="clipped_l2norm")
    p.add_argument("--lambda_softmax_CMRC", default=4.0, type=float)
    p.add_argument("--hid_router", type=int, default=768)
    # --- extensions
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "f32"])
    p.add_argument("--num_classes", default=3, type=int)
    p.add_argument("--num_cells", default=6, type=int, help="cells per routing layer: the first n of ric, glac, imrc, cmrc, "
                   "crcmc, gesc (6 = the reference; 2..5 = declared-subset extension, BASELINE configs[4] uses 4)")
    p.add_argument("--image_size", default=224, type=int)
    p.add_argument("--patch_size", default=32, type=int)
    p.add_argument("--encoder_layers", default=12, type=int)
    p.add_argument("--bert_dropout", default=0.1, type=float,
                   help="hidden_dropout_prob = attention_probs_dropout_prob of the BERT config (bert-base default 0.1)")
    p.add_argument("--train_samples", default=512, type=int)
    p.add_argument("--eval_samples", default=128, type=int)
    p.add_argument("--num_workers", default=4, type=int)
    p.add_argument("--dp_overlap", action="store_true")
    p.add_argument("--dp_grad_comm", default="f32", choices=["f32", "bf16"], help="dtype of the gradient buckets on the links")
    p.add_argument("--dp_shard_optimizer", action="store_true",
                   help="per bucket: reduce-scatter of the gradients, AdamW on this rank's stripe, all-gather of the updated weights "
                        "(combines with --dp_overlap; the RCCL reduce-scatter / all-gather calls are rehearsed with gloo only so far)")
    p.add_argument("--dp_algorithm", default="all_reduce", choices=["all_reduce", "reduce_scatter_all_gather"],
                   help="gradient reduction per bucket: RCCL's all-reduce, or its two phases issued explicitly")
    p.add_argument("--dp_exact", action="store_true", help="data parallelism: BatchNorm statistics of the GLAC cells, the [B,B] similarity "
                   "matrices and the JS loss over the GLOBAL batch (the reference's single-GPU semantics) instead of per rank")
    p.add_argument("--cleanup_output", action="store_true", help="reference behaviour: rmtree('./output') at the end")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    from .config import TextConfig, VisionConfig
    from .data import SyntheticMSDDataset, make_loader
    from .dp import init_process_group_from_env
    from .modules import UnimoModelF
    from .train import MSDTrainer

    rank, world = init_process_group_from_env()
    if args.device == "cuda":
        args.device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
    args.compute_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}[args.dtype]
    set_seed(args.seed)
    if args.save_path is not None and rank == 0:
        os.makedirs(args.save_path, exist_ok=True)
    logger.info(args)
    if args.batch_size % world:
        raise SystemExit(f"--batch_size {args.batch_size} must be divisible by the world size {world}")
    per_rank = args.batch_size // world
    ntok = (args.image_size // args.patch_size) ** 2 + 1

    def loader(n, seed, shuffle):
        ds = SyntheticMSDDataset(n, args.max_seq, args.image_size, args.num_classes, seed=seed, num_image_tokens=ntok)
        sampler = None
        if world > 1 and shuffle:  # only the TRAINING set is sharded; every rank evaluates the whole dev / test set
            sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle,
                                                                      seed=args.seed)
        return make_loader(ds, per_rank, shuffle, args.num_workers, drop_last=shuffle, sampler=sampler)

    train_dl, dev_dl, test_dl = loader(args.train_samples, 1, True), loader(args.eval_samples, 2, False), loader(args.eval_samples, 3, False)
    text_config = TextConfig(num_hidden_layers=args.encoder_layers, hidden_dropout_prob=args.bert_dropout,
                             attention_probs_dropout_prob=args.bert_dropout)
    vision_config = VisionConfig(num_hidden_layers=args.encoder_layers, image_size=args.image_size, patch_size=args.patch_size)
    model = UnimoModelF(args=args, vision_config=vision_config, text_config=text_config, num_classes=args.num_classes)
    trainer = MSDTrainer(train_data=train_dl, dev_data=dev_dl, test_data=test_dl, model=model, args=args, logger=logger,
                         writer=None)
    trainer.train(None, None)  # pretrained CLIP/BERT dicts are unavailable offline; ingest is exercised in tests
    if trainer.samples_per_sec:
        logger.info("training throughput: %.1f samples/s on %d GPU(s)", trainer.samples_per_sec, world)


if __name__ == "__main__":
    main()
