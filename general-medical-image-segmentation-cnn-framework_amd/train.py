"""``python train.py config=unet config.key=value`` -- the reference's training entry point (train.py:90-388)
on the MI355X hot path: same config keys, registry names, init policy, optimizer / StepLR, step semantics
(train.py:187-221 via engine.train_step) and checkpoint format
(``{"model", "optim", "scheduler", "epoch"}`` in latest_checkpoint.pt / checkpoint_%04d.pt, train.py:285-306;
resume when load_mode == 1, train.py:123-140).  TensorBoard / rich / accelerate are replaced by a JSONL scalar
log, the logging module and mi355seg.distributed."""
import json
import logging
import os
import sys
import time

import torch
from torch.optim.lr_scheduler import StepLR

from . import distributed as D
from .config import compose, parse_patch_size
from .data import make_loader
from .utils.metric import metric_from_counts
from .engine import GraphedTrainStep, make_adam, mixed_precision_dtype, train_step, weights_init_normal
from .registry import build_model


class AverageMeter:
    def __init__(self):
        self.val = self.sum = self.count = self.avg = 0.0

    def update(self, v, n=1):
        self.val = v
        self.sum += v * n
        self.count += n
        self.avg = self.sum / max(self.count, 1)


def get_logger(config):
    os.makedirs(config.hydra_path, exist_ok=True)
    log = logging.getLogger("mi355seg.train")
    log.setLevel(logging.DEBUG)
    log.handlers.clear()
    log.addHandler(logging.StreamHandler(sys.stdout))
    log.addHandler(logging.FileHandler(os.path.join(config.hydra_path, f"{config.job_name}.log")))
    log.propagate = False
    return log


def train(config, model, logger):
    rank, world, local = D.init_from_env()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    model = model.to(device)
    # config.hip_graph=true: the iteration is captured once and replayed -- for launch-bound shapes (64^3 patches, UNETR's token
    # path); the learning rate then lives in a device tensor so that StepLR still reaches the captured step.  Data parallel: two
    # graphs (through backward, optimizer step) with the gradient reducer between their replays.
    # config.capturable_adam=true: the same device-resident optimizer state without the graph (eager runs that must match a graphed one)
    flag = lambda name: str(config.get(name) if hasattr(config, "get") else getattr(config, name, False)).lower() in ("1", "true", "yes")
    hip_graph = flag("hip_graph")
    optimizer = make_adam(model.parameters(), lr=torch.tensor(float(config.init_lr), device=device), capturable=True) if (hip_graph or flag("capturable_adam")) \
        else make_adam(model.parameters(), lr=config.init_lr)                   # train.py:109 (torch's fused single-kernel Adam on the GPU)
    scheduler = StepLR(optimizer, step_size=config.scheduler_step_size, gamma=config.scheduler_gamma) \
        if config.use_scheduler else None                                       # train.py:119-120
    elapsed_epochs = 0
    if config.load_mode == 1:                                                   # train.py:123-140
        logger.info(f"load model from: {config.ckpt}")
        ckpt = torch.load(config.ckpt, map_location="cpu")
        state = {k[len("module."):] if k.startswith("module.") else k: v for k, v in ckpt["model"].items()}
        model.load_state_dict(state)                                            # accepts the reference's DDP-prefixed keys
        optimizer.load_state_dict(ckpt["optim"])
        if scheduler is not None and ckpt.get("scheduler") is not None:
            scheduler.load_state_dict(ckpt["scheduler"])
        elapsed_epochs = ckpt["epoch"]
    model.train()
    scalars = open(os.path.join(config.hydra_path, "scalars.jsonl"), "a") if rank == 0 else None
    loader = make_loader(config, device, config.in_classes, seed=1234 + rank)
    # accelerator.prepare (train.py:167-169): rank 0's parameters / buffers everywhere (after the optional checkpoint
    # load, so a resumed rank 0 wins), flat buffers for the per-step broadcast, bucketed gradient reducer
    reducer = D.setup_replica(model)
    act_dtype = mixed_precision_dtype(config)                                   # Accelerator(mixed_precision=...) of train.py:167
    if act_dtype != torch.float32:
        logger.info(f"mixed precision: activations in {act_dtype} (fp32 parameters, gradients, statistics, loss)")
    epochs = config.epochs - elapsed_epochs
    graphed = None
    iteration = elapsed_epochs * len(loader)
    loss_meter, dice_meter = AverageMeter(), AverageMeter()
    for epoch in range(elapsed_epochs + 1, elapsed_epochs + epochs + 1):
        t_epoch = time.time()
        for i, batch in enumerate(loader):
            t0 = time.time()
            x, gt = batch["source"]["data"], batch["gt"]["data"]
            if world > 1:
                D.broadcast_buffers(model, async_op=True)               # launched here, waited for at the forward's first norm layer
            if hip_graph and graphed is None:
                # the first iteration runs inside the constructor (eager, on a side stream: workspaces sized, kernel attributes set),
                # then the iteration is captured for the following ones; a replay waits for the buffer broadcast launched above
                graphed = GraphedTrainStep(model, optimizer, x, gt, warmup=1, dtype=act_dtype, grad_hook=reducer)
                out = dict(graphed.first)
                out["jaccard"], out["dice"] = metric_from_counts(out["counts"].cpu().tolist())
            elif hip_graph:
                out = graphed(x, gt)
            else:
                out = train_step(model, optimizer, x, gt, grad_hook=reducer, dtype=act_dtype)    # train.py:187-221
            iteration += 1
            loss_meter.update(out["loss"].item(), x.size(0))
            dice_meter.update(out["dice"], x.size(0))
            if scalars:
                scalars.write(json.dumps({"iteration": iteration, "Training/Loss": loss_meter.val, "Training/dice": dice_meter.val}) + "\n")
                scalars.flush()
            logger.info(f"\nEpoch: {epoch} Batch: {i}, train time: {time.time() - t0:.3f}s\nLoss: {loss_meter.val}\nDice: {dice_meter.val}\n")
        if scheduler is not None:
            scheduler.step()
            logger.info(f"Learning rate:  {scheduler.get_last_lr()[0]}")
        logger.info(f"\nEpoch {epoch} used time:  {time.time() - t_epoch:.3f} s\nLoss Avg:  {loss_meter.avg}\nDice Avg:  {dice_meter.avg}\n")
        if rank == 0:
            state = {"model": model.state_dict(), "optim": optimizer.state_dict(),
                     "scheduler": scheduler.state_dict() if scheduler is not None else None, "epoch": epoch}
            torch.save(state, os.path.join(config.hydra_path, config.latest_checkpoint_file))
            if epoch % config.epochs_per_checkpoint == 0:
                torch.save(state, os.path.join(config.hydra_path, f"checkpoint_{epoch:04d}.pt"))
    if scalars:
        scalars.close()
    return {"loss_avg": loss_meter.avg, "dice_avg": dice_meter.avg, "epoch": elapsed_epochs + epochs}


def main(argv=None, conf_dir=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    conf_dir = conf_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conf")
    config = compose(conf_dir, argv, job_name="train")
    parse_patch_size(config)
    # ``accelerate launch --num_processes N train.py`` of the reference: ``config.gpus=N`` (or MI355SEG_GPUS=N) starts the N
    # ranks from here when no launcher did (WORLD_SIZE unset) -- before this process has made any GPU call -- and waits
    gpus = int(config.get("gpus", os.environ.get("MI355SEG_GPUS", 1)) or 1)
    if gpus > 1 and "WORLD_SIZE" not in os.environ:
        # always the repo-level wrapper (it puts the package on the path; this file uses relative imports and whatever
        # sys.argv[0] is -- pytest, ``python -c``, another driver script -- is not the train entry point)
        script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "train.py")
        rc = D.self_launch(gpus, [script] + argv)
        if rc:
            raise SystemExit(rc)
        return config, None
    model = build_model(config)                          # train.py:324-373
    model.apply(weights_init_normal(config.init_type))   # train.py:374
    logger = get_logger(config)
    logger.info("\nParameter Settings:\n" + "".join(f"{k}: {v}\n" for k, v in config.items()))
    result = train(config, model, logger)
    logger.info(f"scalar log saved in:{config.hydra_path}")
    return config, result


if __name__ == "__main__":
    main()
