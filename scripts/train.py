#!/usr/bin/env python3
"""Train a Style-SeqCVAE captioner on MI355X - counterpart of the reference's var_updown/scripts/train.py:26-188 with
the same flags, config keys, seeds, optimiser (SGD momentum/weight-decay, LambdaLR linear decay), decoder-LSTM freeze
schedule, clip_grad_norm, scalar names and checkpoint layout ({"model": state_dict, "optimizer": ...}).

One process per GPU: `python scripts/train.py --config cfg.yaml --gpu-ids 0` or, for N GPUs,
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/train.py ... --gpu-ids 0 1 ...`
(the reference's nn.DataParallel path crashes for training: SURVEY §2.1).  OPTIM.BATCH_SIZE is the GLOBAL batch.
Data: --train-tensors file.pt (see ssc_runtime/data.py) or --synthetic N (random features / captions).
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssc_runtime.config import Config  # noqa: E402
from ssc_runtime.data import SyntheticCaptionData, TensorFileData, cycle  # noqa: E402
from ssc_runtime.vocab import Vocabulary  # noqa: E402
from var_updown.models import UpDownCaptioner  # noqa: E402

parser = argparse.ArgumentParser("Train a Style-SeqCVAE UpDown captioner (MI355X).")
parser.add_argument("--config", required=True)
parser.add_argument("--config-override", default=[], nargs="*")
parser.add_argument("--gpu-ids", required=True, nargs="+", type=int)
parser.add_argument("--cpu-workers", type=int, default=0)
parser.add_argument("--in-memory", action="store_true")
parser.add_argument("--skip-validation", action="store_true")
parser.add_argument("--serialization-dir", default="checkpoints/experiment")
parser.add_argument("--checkpoint-every", default=10000, type=int)
parser.add_argument("--start-from-checkpoint", default="")
parser.add_argument("--train-tensors", default="", help=".pt file with image_features / caption_tokens / sentiment")
parser.add_argument("--synthetic", type=int, default=0, help="train on N synthetic images (BASELINE.md §4)")
parser.add_argument("--vocab-size", type=int, default=10000, help="vocabulary size for --synthetic")
parser.add_argument("--num-boxes", type=int, default=36)
parser.add_argument("--eps-source", default="device", choices=["cpu", "device"],
                    help="cpu: the reference's CPU randn stream per step; device: GPU RNG, no host traffic")
parser.add_argument("--fused-optimizer", action="store_true",
                    help="clip + SGD in one HIP pass on the flat buffers instead of torch.optim.SGD")


def main():
    _A = parser.parse_args()
    _C = Config(_A.config, _A.config_override)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if -1 in _A.gpu_ids:
        raise SystemExit("--gpu-ids -1 (CPU) is not available: this build has no CPU path")
    gpu = _A.gpu_ids[local % len(_A.gpu_ids)]
    torch.cuda.set_device(gpu)
    device = torch.device("cuda", gpu)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    if rank == 0:
        print(_C)
        os.makedirs(_A.serialization_dir, exist_ok=True)
        _C.dump(os.path.join(_A.serialization_dir, "config.yml"))

    random.seed(_C.RANDOM_SEED)
    np.random.seed(_C.RANDOM_SEED)
    torch.manual_seed(_C.RANDOM_SEED)

    if _A.synthetic:
        vocabulary = Vocabulary.synthetic(_A.vocab_size)
        data = SyntheticCaptionData(_A.synthetic, _A.num_boxes, _C.MODEL.IMAGE_FEATURE_SIZE, _C.DATA.MAX_CAPTION_LENGTH,
                                    _A.vocab_size, seed=1234)
    else:
        vocabulary = Vocabulary.from_files(_C.DATA.VOCABULARY)
        if not _A.train_tensors:
            raise SystemExit("the h5/nltk dataset readers are out of scope: pass --train-tensors file.pt or --synthetic N")
        data = TensorFileData(_A.train_tensors)
    if _C.OPTIM.BATCH_SIZE % world:
        raise SystemExit("OPTIM.BATCH_SIZE (global) must be divisible by the number of ranks")
    loader = cycle(data, _C.OPTIM.BATCH_SIZE // world, device, rank, world, seed=_C.RANDOM_SEED)

    model = UpDownCaptioner.from_config(_C, vocabulary=vocabulary, cbs_simple=_C.MODEL.CBS_SIMPLE, device=device).to(device)
    model.eps_source = _A.eps_source
    model.train()
    eng = model._engine()
    optimizer = torch.optim.SGD(model.parameters(), lr=_C.OPTIM.LR, momentum=_C.OPTIM.MOMENTUM,
                                weight_decay=_C.OPTIM.WEIGHT_DECAY)
    lr_scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda it: 1 - it / _C.OPTIM.NUM_ITERATIONS)
    start_iteration = 1
    if _A.start_from_checkpoint:
        ckpt = torch.load(_A.start_from_checkpoint, map_location=device, weights_only=True)
        model.load_state_dict(ckpt["model"])
        if "optimizer" in ckpt and not _A.fused_optimizer:
            optimizer.load_state_dict(ckpt["optimizer"])
        start_iteration = int(ckpt.get("iteration", 0)) + 1  # correct resume (the reference restarts at 1: train.py:149)
    log = open(os.path.join(_A.serialization_dir, "scalars.jsonl"), "a") if rank == 0 else None

    t0 = time.time()
    for iteration in range(start_iteration, _C.OPTIM.NUM_ITERATIONS + 1):
        train_decoder = (iteration > _C.OPTIM.EPOCH_START_DECODER_TRAINING
                         or iteration % _C.OPTIM.BEFORE_UPDATE_DECODER_EVERY == 0)   # train.py:156-161
        for p in model._updown_cell._language_lstm_cell_decoder.parameters():
            p.requires_grad = train_decoder
        batch = next(loader)
        lr = _C.OPTIM.LR * (1 - (iteration - 1) / _C.OPTIM.NUM_ITERATIONS)
        if _A.fused_optimizer:
            B, L = batch["caption_tokens"].shape
            eps = model._draw_eps(L + 1, B, device)
            loss_b, kld_b = eng.train_step(batch["image_features"], batch["caption_tokens"], batch["sentiment"], eps, lr=lr,
                                           kld_weight=_C.MODEL.KLD_WEIGHT, momentum=_C.OPTIM.MOMENTUM,
                                           weight_decay=_C.OPTIM.WEIGHT_DECAY, max_norm=_C.OPTIM.CLIP_GRADIENTS,
                                           decoder_frozen=not train_decoder)
            reconstr_loss, kld_loss = loss_b.mean(), kld_b.mean()
            loss = reconstr_loss + kld_loss / _C.MODEL.KLD_WEIGHT
        else:
            optimizer.zero_grad()
            out = model(batch["image_features"], None, None, batch["caption_tokens"], batch["sentiment"])
            reconstr_loss, kld_loss = out["loss"].mean(), out["kld"].mean()
            loss = reconstr_loss + kld_loss / _C.MODEL.KLD_WEIGHT
            loss.backward()
            if world > 1:  # mean over ranks of the local-mean gradients
                import torch.distributed as dist
                for p in model.parameters():
                    if p.grad is not None:
                        dist.all_reduce(p.grad)
                        p.grad.div_(world)
            torch.nn.utils.clip_grad_norm_(model.parameters(), _C.OPTIM.CLIP_GRADIENTS)
            optimizer.step()
            lr_scheduler.step()
            lr = optimizer.param_groups[0]["lr"]
        if rank == 0 and (iteration % 100 == 0 or iteration == start_iteration):
            rec = {"iteration": iteration, "1reconstr_loss": float(reconstr_loss), "2kld_loss": float(kld_loss),
                   "3loss": float(loss), "4learning_rate": lr, "elapsed_s": time.time() - t0}
            log.write(json.dumps(rec) + "\n")
            log.flush()
            if iteration % 2000 == 0 or iteration == start_iteration:
                print("{:6f}    {:6f}    {:6f}".format(rec["3loss"], rec["1reconstr_loss"], rec["2kld_loss"]))
        if rank == 0 and iteration % _A.checkpoint_every == 0:
            sd = {"model": model.state_dict(), "iteration": iteration}
            if not _A.fused_optimizer:
                sd["optimizer"] = optimizer.state_dict()
            torch.save(sd, os.path.join(_A.serialization_dir, f"checkpoint_{iteration}.pth"))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
