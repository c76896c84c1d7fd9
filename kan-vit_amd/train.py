"""Training entry point -- drop-in for the reference's train.py (same flags and defaults,
train.py:87-97; same step order forward -> CE loss -> zero_grad -> backward -> Adam.step,
train.py:31-40), running on the MI355X kernels, with optional pure data parallelism:

    python train.py --model-type cheby                                 # 1 GPU, CIFAR-100 if torchvision + data exist
    python train.py --model-type cheby --synthetic --epochs 1          # synthetic stream, no dataset needed
    python train.py --model-type cheby --synthetic --graph             # the whole step replayed from one HIP graph
    python train.py --model-type fast --synthetic --amp bf16           # bf16 autocast + kanvit kernels on the bf16 matrix cores
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --dp --synthetic ...

Additions over the reference (none change a default): --synthetic / --image-size / --in-chans /
--n-patches / --out-d (geometry), --seed, --dp, --steps-per-epoch, --amp, --graph, --no-tuned-gemms,
--no-step-metrics (the per-step metrics stay on the device either way: one host sync per epoch instead of the
reference's three per step, train.py:37,42-44).

`main(args, batches=None, init_state=None)` also serves the parity tests: `batches` (a list of (x, y) tensors) replaces
the data stream, `init_state` (a reference-layout state dict) replaces the random initialisation, and the returned dict
carries the per-step loss trajectory."""
import argparse
import logging
import os

import torch
from torch.optim import Adam

from kanvit import dp as kdp
from model import VisionTransformer
from utils import calculate_metrics, save_metrics, setup_logging


def synthetic_loader(n_steps, batch, chw, out_d, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    for _ in range(n_steps):
        yield (torch.randn(batch, *chw, device=device, generator=g),
               torch.randint(0, out_d, (batch,), device=device, generator=g))


def cifar_loaders(batch, rank, world):
    """CIFAR-100 with the reference's transforms (train.py:100-118); built ONCE.  Returns (train, test, sampler)."""
    from torchvision import transforms
    from torchvision.datasets import CIFAR100
    norm = transforms.Normalize(mean=[0.5071, 0.4867, 0.4408], std=[0.2675, 0.2565, 0.2761])
    tr = transforms.Compose([transforms.RandomHorizontalFlip(), transforms.RandomCrop(32, padding=4),
                             transforms.ToTensor(), norm])
    te = transforms.Compose([transforms.ToTensor(), norm])
    train = CIFAR100(root='./cifar100', train=True, download=True, transform=tr)
    test = CIFAR100(root='./cifar100', train=False, download=True, transform=te)
    sampler = torch.utils.data.distributed.DistributedSampler(train, world, rank, shuffle=True) if world > 1 else None
    mk = torch.utils.data.DataLoader
    return (mk(train, batch_size=batch, shuffle=sampler is None, sampler=sampler, num_workers=8, pin_memory=True),
            mk(test, batch_size=batch, shuffle=False, num_workers=8, pin_memory=True), sampler)


class _GraphedStep:
    """The whole train step (forward, loss, zero_grad, backward, Adam) captured once in a HIP graph and replayed per batch:
    the launch-bound geometries (train.py's own defaults: 64-wide model, 17 tokens) spend their time in ~1000 launches of
    microsecond kernels.  Batches are copied into static buffers; a batch of another shape (the last, short one) runs eagerly.

    Capturing needs lazily created state (optimizer moments, library workspaces) to exist, so one throw-away step runs on a
    side stream first; parameters are then restored and the Adam moments / step counters zeroed, which is exactly the state
    of a fresh optimizer -- the replayed trajectory equals the eager one."""

    def __init__(self, eager_step, model, optimizer, x, y):
        self.eager = eager_step
        self.sx, self.sy = x.clone(), y.clone()
        saved = [p.detach().clone() for p in model.parameters()]
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):
            eager_step(self.sx, self.sy)
        torch.cuda.current_stream(x.device).wait_stream(side)
        with torch.no_grad():
            for p, s in zip(model.parameters(), saved):
                if p.requires_grad:             # only what the throw-away step can have changed: rewriting FastKAN's frozen
                    p.copy_(s)                  # rbf.grid would bump its version and re-derive grid facts (host syncs) in capture
            for st in optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.logits = eager_step(self.sx, self.sy)

    def __call__(self, x, y):
        if x.shape != self.sx.shape:
            return self.eager(x, y)
        self.sx.copy_(x, non_blocking=True)
        self.sy.copy_(y, non_blocking=True)
        self.graph.replay()
        return self.loss, self.logits


def main(args, batches=None, init_state=None):
    rank, world, local = 0, 1, 0
    if args.dp:
        import torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        local = int(os.environ.get("LOCAL_RANK", rank))
        # rehearsal hooks for a ONE-GPU box (same as bench.py's; never set by a real launch): every rank on cuda:0, gloo transport
        if os.environ.get("KANVIT_SHARE_GPU") == "1":
            local = 0
        torch.cuda.set_device(local)
        backend = os.environ.get("KANVIT_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        if args.batch_size % world:
            raise SystemExit(f"--batch-size {args.batch_size} is not divisible by the {world} ranks (equal shards keep the "
                             "all-reduced mean equal to the global-batch mean)")
        if args.graph:
            raise SystemExit("--graph captures a single-GPU step; it is not combined with --dp")
    device = torch.device(args.device if not args.dp else f"cuda:{local}")
    if device.type != "cuda":
        raise SystemExit("this build runs the hot path on the MI355X only; --device must be a cuda device")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    torch.cuda.set_device(device)           # the kernels launch on the model's device also with --device cuda:N
    torch.manual_seed(args.seed)
    chw = (args.in_chans, args.image_size, args.image_size)
    model = VisionTransformer(chw, n_patches=args.n_patches, n_blocks=args.n_blocks, d_hidden=args.d_hidden,
                              n_heads=args.n_heads, out_d=args.out_d, type=args.model_type)
    if init_state is not None:
        model.load_state_dict(init_state)
    model = model.to(device)
    kdp.broadcast_parameters(model)
    criterion = torch.nn.CrossEntropyLoss()
    if not args.no_tuned_gemms:
        from kanvit import tuned
        tuned.enable_tuned_gemms()          # recorded kernel selections for the stock FF GEMMs (no run-time tuning)
    # fused=True: same update rule as the reference's Adam, one multi-tensor kernel per parameter chunk instead of ~8 per step
    optimizer = Adam(model.parameters(), lr=args.learning_rate, fused=True, capturable=bool(args.graph))
    reducer = kdp.GradReducer(model.parameters()) if world > 1 else None
    metrics_file = setup_logging(args.log_dir) if rank == 0 else None
    logging.info(f"Using device: {device} ({torch.cuda.get_device_name(device)}), world {world}, amp {args.amp}, "
                 f"graph {bool(args.graph)}")
    amp = args.amp == "bf16"

    def eager_step(x, y):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            y_hat = model(x)
            loss = criterion(y_hat.float(), y)
        if reducer is not None:
            reducer.zero_grad()
            reducer.scale_loss(loss).backward()     # mean over ranks inside the collective (RCCL AVG) or the loss: kanvit/dp.py
            reducer.finish()
        else:
            optimizer.zero_grad()
            loss.backward()
        optimizer.step()
        return loss.detach(), y_hat.detach()

    step = eager_step
    per_rank = args.batch_size // world if args.dp else args.batch_size
    train_loader = test_loader = sampler = None
    if batches is None and not args.synthetic:
        train_loader, test_loader, sampler = cifar_loaders(per_rank, rank, world)
    history = {"losses": [], "epoch_loss": []}
    model.train()
    for epoch in range(args.epochs):
        if batches is not None:
            loader, n_batches = batches, len(batches)
        elif args.synthetic:
            loader = synthetic_loader(args.steps_per_epoch, per_rank, chw, args.out_d, device, args.seed + 1000 * epoch + rank)
            n_batches = args.steps_per_epoch
        else:
            if sampler is not None:
                sampler.set_epoch(epoch)            # a different shuffle per epoch, the same on every rank
            loader, n_batches = train_loader, len(train_loader)
        step_losses, ys, preds, probs = [], [], [], []
        for x, y in loader:
            x, y = x.to(device, non_blocking=True), y.to(device, non_blocking=True)
            if args.graph and step is eager_step:
                step = _GraphedStep(eager_step, model, optimizer, x, y)
            loss, y_hat = step(x, y)
            step_losses.append(loss.clone() if args.graph else loss)
            if not args.no_step_metrics:
                ys.append(y)
                preds.append(y_hat.argmax(dim=1))
                probs.append(torch.softmax(y_hat.float(), dim=1))
        losses = torch.stack(step_losses) if step_losses else torch.zeros(0, device=device)
        if world > 1:                           # every rank's step loss is the mean over ITS shard; equal shards -> the mean
            torch.distributed.all_reduce(losses)        # of rank means is the global-batch mean the reference would log
            losses /= world
        losses = losses.cpu()                   # the epoch's one host sync
        history["losses"] += [float(v) for v in losses]
        train_loss = float(losses.sum() / max(n_batches, 1))
        history["epoch_loss"].append(train_loss)
        if rank == 0:
            logging.info(f"Epoch {epoch + 1}/{args.epochs}\n  Train Loss: {train_loss:.4f}")
            if ys:
                acc, bal, f1, auc = calculate_metrics(torch.cat(ys).cpu().numpy(), torch.cat(preds).cpu().numpy(),
                                                      torch.cat(probs).cpu().numpy(), num_classes=args.out_d)
                logging.info(f"  Train Accuracy: {acc:.4f}\n  Train Balanced Accuracy: {bal:.4f}\n"
                             f"  Train F1 Score: {f1:.4f}\n  Train ROC AUC: {auc:.4f}")
                if epoch == args.epochs - 1:
                    save_metrics(metrics_file, epoch + 1, "Train", train_loss, acc, bal, f1, auc, flag=0)
                    history["metrics_file"] = metrics_file

    if test_loader is not None and rank == 0:
        model.eval()
        with torch.no_grad():
            tl, ys, preds, probs = [], [], [], []
            for x, y in test_loader:
                x, y = x.to(device), y.to(device)
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    y_hat = model(x)
                tl.append(criterion(y_hat.float(), y))
                ys.append(y), preds.append(y_hat.argmax(dim=1)), probs.append(torch.softmax(y_hat.float(), dim=1))
            test_loss = float(torch.stack(tl).sum() / len(test_loader))
            acc, bal, f1, auc = calculate_metrics(torch.cat(ys).cpu().numpy(), torch.cat(preds).cpu().numpy(),
                                                  torch.cat(probs).cpu().numpy(), num_classes=args.out_d)
            logging.info(f"Test Results:\n  Test Loss: {test_loss:.4f}\n  Test Accuracy: {acc:.4f}")
            save_metrics(metrics_file, args.epochs, "Test", test_loss, acc, bal, f1, auc, flag=1)
    if args.dp:
        torch.distributed.barrier()             # the other ranks wait for rank 0's evaluation before the group goes away
        torch.distributed.destroy_process_group()
    history["model"] = model
    return history


def parse(argv=None):
    p = argparse.ArgumentParser(description='Benchmark Vision Transformer on CIFAR-100')
    # the reference's flags, verbatim (train.py:87-97)
    p.add_argument('--epochs', type=int, default=20, help='number of epochs to train')
    p.add_argument('--batch-size', type=int, default=128, help='batch size for training')
    p.add_argument('--learning-rate', type=float, default=0.001, help='learning rate for optimizer')
    p.add_argument('--device', type=str, default='cuda' if torch.cuda.is_available() else 'cpu',
                   help='device to use for training (cuda/cpu)')
    p.add_argument('--model-type', type=str, default='vanilla', help='variant to run')
    p.add_argument('--n-blocks', type=int, default=8, help='number of transformer blocks')
    p.add_argument('--d-hidden', type=int, default=64, help='hidden dimension of transformer block')
    p.add_argument('--n-heads', type=int, default=8, help='number of attention heads')
    p.add_argument('--log-dir', type=str, default='logs', help='directory to store logs')
    # additions (defaults reproduce the geometry hard-coded at train.py:19)
    p.add_argument('--synthetic', action='store_true', help='synthetic image stream instead of CIFAR-100')
    p.add_argument('--steps-per-epoch', type=int, default=100, help='steps per epoch of the synthetic stream')
    p.add_argument('--image-size', type=int, default=32)
    p.add_argument('--in-chans', type=int, default=3)
    p.add_argument('--n-patches', type=int, default=4)
    p.add_argument('--out-d', type=int, default=100)
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--dp', action='store_true', help='data parallel: one process per GPU (launch with torch.distributed.run)')
    p.add_argument('--no-step-metrics', action='store_true', help='skip per-step metric accumulation')
    p.add_argument('--amp', choices=['off', 'bf16'], default='off',
                   help='bf16: autocast for the stock dense ops and the kanvit kernels on the bf16 matrix cores (fp32 I/O and '
                        'accumulation); off (default): the reference\'s fp32 arithmetic')
    p.add_argument('--graph', action='store_true', help='capture the whole train step in a HIP graph and replay it per batch')
    p.add_argument('--no-tuned-gemms', action='store_true', help='library-default kernel selection for the stock GEMMs')
    return p.parse_args(argv)


if __name__ == "__main__":
    main(parse())
