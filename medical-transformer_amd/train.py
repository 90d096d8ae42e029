#!/usr/bin/env python
"""Drop-in for the reference's train.py: same flags, defaults, stdout lines and checkpoint paths
(reference train.py:30-66, 93-117, 126-217), driving the MI355X-native model.

Differences, all host-side plumbing:
  * data loading uses PIL/numpy (utils.py here) -- cv2/torchvision/skimage are not in the MI355X image;
  * --device is honoured (the reference hard-codes "cuda", train.py:93); the kernels need a GPU;
  * multi-GPU = one process per GPU under torchrun (torch.distributed, RCCL) with a single flat-bucket
    gradient all-reduce per step, instead of nn.DataParallel(device_ids=[0,1]) (train.py:104-107);
    --batch_size stays the GLOBAL batch and is split over the ranks, as DataParallel splits it;
  * batches are decoded, pinned and copied to the GPU one step ahead (medt_amd.data.DevicePrefetcher);
  * the step runs as a replayed hipGraph with a fused flat Adam (medt_amd.trainer); --eager disables it;
  * --synthetic N writes N synthetic PNG pairs into --train_dataset first (BASELINE.json config 1 plumbing);
  * the per-step threshold-and-copy-to-host of the output (train.py:142-152) is dropped: its result is unused.
"""
import argparse
import os

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

import lib
from metrics import LogNLLLoss
from medt_amd import dp
from medt_amd.data import DevicePrefetcher, imwrite, make_synthetic_dataset
from medt_amd.optim import FlatAdam
from medt_amd.trainer import InferStep, TrainStep

parser = argparse.ArgumentParser(description='MedT')
parser.add_argument('-j', '--workers', default=16, type=int, metavar='N', help='number of data loading workers (default: 8)')
parser.add_argument('--epochs', default=400, type=int, metavar='N', help='number of total epochs to run(default: 400)')
parser.add_argument('--start-epoch', default=0, type=int, metavar='N', help='manual epoch number (useful on restarts)')
parser.add_argument('-b', '--batch_size', default=1, type=int, metavar='N', help='batch size (default: 1)')
parser.add_argument('--learning_rate', default=1e-3, type=float, metavar='LR', help='initial learning rate (default: 0.001)')
parser.add_argument('--momentum', default=0.9, type=float, metavar='M', help='momentum')
parser.add_argument('--weight-decay', '--wd', default=1e-5, type=float, metavar='W', help='weight decay (default: 1e-5)')
parser.add_argument('--train_dataset', required=True, type=str)
parser.add_argument('--val_dataset', type=str)
parser.add_argument('--save_freq', type=int, default=10)
parser.add_argument('--modelname', default='MedT', type=str, help='type of model')
parser.add_argument('--cuda', default="on", type=str, help='switch on/off cuda option (default: off)')
parser.add_argument('--aug', default='off', type=str, help='turn on img augmentation (default: False)')
parser.add_argument('--load', default='default', type=str, help='load a pretrained model')
parser.add_argument('--save', default='default', type=str, help='save the model')
parser.add_argument('--direc', default='./medt', type=str, help='directory to save')
parser.add_argument('--crop', type=int, default=None)
parser.add_argument('--imgsize', type=int, default=None)
parser.add_argument('--device', default='cuda', type=str)
parser.add_argument('--gray', default='no', type=str)
parser.add_argument('--synthetic', type=int, default=0, help='write this many synthetic PNG pairs into the dataset dirs first')
parser.add_argument('--eager', action='store_true', help='no hipGraph replay')


def main():
    args = parser.parse_args()
    direc, modelname, imgsize = args.direc, args.modelname, args.imgsize
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if args.gray == "yes":
        from utils_gray import JointTransform2D, ImageToImage2D, Image2D
        imgchant = 1
    else:
        from utils import JointTransform2D, ImageToImage2D, Image2D
        imgchant = 3
    device = torch.device(args.device)
    if device.type == "cuda":
        if device.index is None:
            device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl" if device.type == "cuda" else "gloo")
    if args.synthetic:
        # rank 0 writes the PNGs; nobody lists the directories before they are complete
        if rank == 0:
            make_synthetic_dataset(args.train_dataset, args.synthetic, imgsize or 128, 3000, args.gray == "yes")
            if args.val_dataset and args.val_dataset != args.train_dataset:
                make_synthetic_dataset(args.val_dataset, max(2, args.synthetic // 4), imgsize or 128, 3001,
                                       args.gray == "yes")
        if world > 1:
            dist.barrier()

    crop = (args.crop, args.crop) if args.crop is not None else None
    tf_train = JointTransform2D(crop=crop, p_flip=0.5, color_jitter_params=None, long_mask=True)
    tf_val = JointTransform2D(crop=crop, p_flip=0, color_jitter_params=None, long_mask=True)
    train_dataset = ImageToImage2D(args.train_dataset, tf_train)
    val_dataset = ImageToImage2D(args.val_dataset or args.train_dataset, tf_val)
    Image2D(args.val_dataset or args.train_dataset)                    # predict_dataset: constructed, unused (:87)
    if args.batch_size % world:
        raise SystemExit("--batch_size (global) must be divisible by the number of ranks")
    sampler = torch.utils.data.distributed.DistributedSampler(train_dataset, world, rank, shuffle=True, seed=3000) if world > 1 else None
    dataloader = DataLoader(train_dataset, batch_size=args.batch_size // world, shuffle=sampler is None, sampler=sampler)
    valloader = DataLoader(val_dataset, 1, shuffle=True)

    factories = {"axialunet": lib.models.axialunet, "MedT": lib.models.axialnet.MedT,
                 "gatedaxialunet": lib.models.axialnet.gated, "logo": lib.models.axialnet.logo}
    model = factories[modelname](img_size=imgsize, imgchan=imgchant)
    if world > 1 and rank == 0:
        print("Let's use", world, "GPUs!")
    model.to(device)
    dp.broadcast_parameters(model)

    criterion = LogNLLLoss()
    optimizer = FlatAdam(list(model.parameters()), lr=args.learning_rate, weight_decay=1e-5)
    train_step = TrainStep(model, optimizer, criterion, use_graph=not args.eager)
    infer_step = InferStep(model, use_graph=not args.eager)
    if rank == 0:
        print("Total_params: {}".format(sum(p.numel() for p in model.parameters() if p.requires_grad)))

    seed = 3000
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)

    for epoch in range(args.epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        epoch_running_loss, batch_idx = 0.0, -1
        # host decode + pinned staging + H2D run one step ahead on a copy stream (reference: blocking .to(), :134-135)
        for batch_idx, (X_batch, y_batch, *rest) in enumerate(DevicePrefetcher(dataloader, device)):
            X_batch = X_batch.to(device)
            y_batch = y_batch.to(device)
            loss = train_step(X_batch, y_batch)
            epoch_running_loss += loss.item()
            train_step.check_targets()            # mislabelled masks raise, as F.cross_entropy does in the reference
        if rank == 0:
            print('epoch [{}/{}], loss:{:.4f}'.format(epoch, args.epochs, epoch_running_loss / (batch_idx + 1)))

        if epoch == 10:
            for param in model.parameters():
                param.requires_grad = True
        if (epoch % args.save_freq) == 0 and rank == 0:
            fulldir = direc + "/{}/".format(epoch)
            os.makedirs(fulldir, exist_ok=True)
            for batch_idx, (X_batch, y_batch, *rest) in enumerate(valloader):
                image_filename = rest[0][0] if isinstance(rest[0][0], str) else '%s.png' % str(batch_idx + 1).zfill(3)
                # the model stays in train mode here, as in the reference (:174-184): batch statistics, and every forward
                # updates the running statistics; replayed as one hipGraph per image shape (InferStep)
                y_out = infer_step(X_batch.to(device))
                yHaT = (y_out.detach().cpu().numpy() >= 0.5).astype(np.uint8) * 255
                imwrite(fulldir + image_filename, yHaT[0, 1, :, :])
            torch.save(model.state_dict(), fulldir + args.modelname + ".pth")
            torch.save(model.state_dict(), direc + "final_model.pth")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
