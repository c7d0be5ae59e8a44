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

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL across processes (before HIP loads)
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
parser.add_argument("--zero-eps", action="store_true", help="testing: eps = 0 (deterministic z = mean)")
parser.add_argument("--stop-after", type=int, default=0, help="testing: stop after this iteration (NUM_ITERATIONS keeps defining the schedule)")
parser.add_argument("--attribute-table", default="",
                    help="SENTIMENT_VAE 2: json {attribute word: [Z_SPACE floats]} - the table the reference builds from its sentiment-GloVe / "
                         "SentiWordNet files (updown_captioner.py:79-93); only needed when obj_atts arrive as attribute strings")
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
        sv2 = _C.MODEL.SENTIMENT_VAE == 2 and not _C.MODEL.SIMPLE_VAE
        data = SyntheticCaptionData(_A.synthetic, _A.num_boxes, _C.MODEL.IMAGE_FEATURE_SIZE, _C.DATA.MAX_CAPTION_LENGTH,
                                    _A.vocab_size, seed=1234, obj_dim=_C.MODEL.Z_SPACE if sv2 else 0)
    else:
        vocabulary = Vocabulary.from_files(_C.DATA.VOCABULARY)
        if not _A.train_tensors:
            raise SystemExit("the h5 / nltk dataset readers are not built (h5py, nltk are not installable here): pass "
                             "--train-tensors file.pt (ssc_runtime/data.py: dense or ragged region features) or --synthetic N")
        data = TensorFileData(_A.train_tensors)
    if _C.OPTIM.BATCH_SIZE % world:
        raise SystemExit("OPTIM.BATCH_SIZE (global) must be divisible by the number of ranks")

    extra = {}
    if _C.MODEL.SENTIMENT_VAE == 2:
        # the attribute table (word -> Z_SPACE floats) the reference reads from hard-coded pickle paths (updown_captioner.py:79-86).
        # Needed only for obj_atts given as attribute STRINGS; the tensor files / the synthetic source carry the per-region means.
        extra["mean_choice"] = {k: np.asarray(v) for k, v in json.load(open(_A.attribute_table)).items()} if _A.attribute_table else {}
    model = UpDownCaptioner.from_config(_C, vocabulary=vocabulary, cbs_simple=_C.MODEL.CBS_SIMPLE, device=device, **extra).to(device)
    model.eps_source = _A.eps_source
    model.train()
    eng = model._engine()
    optimizer = torch.optim.SGD(model.parameters(), lr=_C.OPTIM.LR, momentum=_C.OPTIM.MOMENTUM,
                                weight_decay=_C.OPTIM.WEIGHT_DECAY)
    # Learning rate: the reference's LambdaLR(1 - it / NUM_ITERATIONS) stepped once per iteration (train.py:132-134,176) gives
    # iteration i the rate LR * (1 - (i - 1) / N).  It is computed from the iteration number on BOTH paths, so a resumed run
    # continues the decay where it stopped (a fresh LambdaLR would restart at LR).
    eng.dp_autograd = world > 1 and not _A.fused_optimizer
    named = list(model.named_parameters())
    start_iteration = 1
    if _A.start_from_checkpoint:
        # Layout {"model": state_dict, "optimizer": SGD state_dict} as written by the reference's CheckpointManager
        # (updown-baseline/updown/utils/checkpointing.py:81-112); the iteration rides inside the optimizer entry
        # (the reference's train.py:143-149 loads every other top-level key into the model).
        ckpt = torch.load(_A.start_from_checkpoint, map_location=device, weights_only=True)
        model.load_state_dict(ckpt["model"])
        osd = ckpt.get("optimizer")
        if osd is not None:
            if _A.fused_optimizer:
                eng.load_optimizer_state_dict(named, osd)
            else:
                optimizer.load_state_dict({"state": osd["state"], "param_groups": osd["param_groups"]})
            start_iteration = int(osd.get("iteration", 0)) + 1   # correct resume (the reference restarts at 1: train.py:149)
    # batch i of a run is a function of (seed, i): a resumed run continues the data order where it stopped
    loader = cycle(data, _C.OPTIM.BATCH_SIZE // world, device, rank, world, seed=_C.RANDOM_SEED, start_batch=start_iteration - 1)
    log = open(os.path.join(_A.serialization_dir, "scalars.jsonl"), "a") if rank == 0 else None

    t0 = time.time()
    for iteration in range(start_iteration, _C.OPTIM.NUM_ITERATIONS + 1):
        train_decoder = (iteration > _C.OPTIM.EPOCH_START_DECODER_TRAINING
                         or iteration % _C.OPTIM.BEFORE_UPDATE_DECODER_EVERY == 0)   # train.py:156-161
        for p in model._updown_cell._language_lstm_cell_decoder.parameters():
            p.requires_grad = train_decoder
        if _A.stop_after and iteration > _A.stop_after:
            break
        batch = next(loader)
        lr = _C.OPTIM.LR * (1 - (iteration - 1) / _C.OPTIM.NUM_ITERATIONS)
        if _A.zero_eps:
            Bz, Lz = batch["caption_tokens"].shape
            model._eps_override = torch.zeros(Lz + 1, Bz, _C.MODEL.Z_SPACE, device=device)
        if _A.fused_optimizer:
            B, L = batch["caption_tokens"].shape
            eps = model._draw_eps(L + 1, B, device)
            loss_b, kld_b = eng.train_step(batch["image_features"], batch["caption_tokens"], batch["sentiment"], eps, lr=lr,
                                           kld_weight=_C.MODEL.KLD_WEIGHT, momentum=_C.OPTIM.MOMENTUM,
                                           weight_decay=_C.OPTIM.WEIGHT_DECAY, max_norm=_C.OPTIM.CLIP_GRADIENTS,
                                           decoder_frozen=not train_decoder, obj_atts=batch.get("obj_atts"))
            reconstr_loss, kld_loss = loss_b.mean(), kld_b.mean()
            loss = reconstr_loss + kld_loss / _C.MODEL.KLD_WEIGHT
        else:
            optimizer.zero_grad()
            out = model(batch["image_features"], batch.get("obj_atts"), None, batch["caption_tokens"], batch["sentiment"])
            reconstr_loss, kld_loss = out["loss"].mean(), out["kld"].mean()
            loss = reconstr_loss + kld_loss / _C.MODEL.KLD_WEIGHT
            for group in optimizer.param_groups:
                group["lr"] = lr
            loss.backward()   # world > 1: the flat gradient buffer is all-reduced once inside backward (eng.dp_autograd)
            torch.nn.utils.clip_grad_norm_(model.parameters(), _C.OPTIM.CLIP_GRADIENTS)
            optimizer.step()
        if rank == 0 and (iteration % 100 == 0 or iteration == start_iteration or _C.OPTIM.NUM_ITERATIONS <= 100):
            rec = {"iteration": iteration, "1reconstr_loss": float(reconstr_loss), "2kld_loss": float(kld_loss),
                   "3loss": float(loss), "4learning_rate": lr, "elapsed_s": time.time() - t0}
            log.write(json.dumps(rec) + "\n")
            log.flush()
            if iteration % 2000 == 0 or iteration == start_iteration:
                print("{:6f}    {:6f}    {:6f}".format(rec["3loss"], rec["1reconstr_loss"], rec["2kld_loss"]))
        if rank == 0 and iteration % _A.checkpoint_every == 0:
            if _A.fused_optimizer:
                osd = eng.optimizer_state_dict(named, lr, _C.OPTIM.MOMENTUM, _C.OPTIM.WEIGHT_DECAY, iteration)
            else:
                osd = optimizer.state_dict()
                osd["iteration"] = iteration
            torch.save({"model": model.state_dict(), "optimizer": osd},
                       os.path.join(_A.serialization_dir, f"checkpoint_{iteration}.pth"))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
