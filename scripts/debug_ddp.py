import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import model as omodel, synth
    from sacv_amd.detection import models, losses
    H, W = 96, 128
    net = models.__dict__['resnet50_retinanet'](num_classes=80)
    net.load_state_dict(omodel.init_retinanet_state(0))
    net = net.cuda().train()
    mode = os.environ.get('MODE', 'late')
    if mode == 'early':
        net.prepare()  # arena (permuted strides) BEFORE DDP construction
    ddp = torch.nn.parallel.DistributedDataParallel(net, gradient_as_bucket_view=os.environ.get('GBV') == '1')
    crit = losses.__dict__['RetinaLoss']()
    images = torch.tensor(synth.images(100 + rank, 2, H, W)).cuda()
    annots = torch.tensor(synth.annotations(200 + rank, 2, H, W, boxes_per_image=8)).cuda()
    with ddp.no_sync():
        sum(crit(ddp(images), annots).values()).backward()
    own = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    arena1 = net.arena.grads.clone()
    for p in net.parameters():
        p.grad = None
    sum(crit(ddp(images), annots).values()).backward()
    arena2 = net.arena.grads.clone()
    print(rank, 'arena grads pass1 == pass2:', torch.equal(arena1, arena2), flush=True)
    n = 'backbone.layer1.0.conv2.layer.0.weight'
    p = dict(net.named_parameters())[n]
    t_perm = own[n].clone(); dist.all_reduce(t_perm)
    t_cont = own[n].contiguous().clone(); dist.all_reduce(t_cont)
    sc = float(t_cont.abs().max())
    print(rank, mode, 'perm-vs-cont allreduce diff', float((t_perm - t_cont).abs().max()) / sc,
          'ddp vs cont/2', float((p.grad - t_cont / 2).abs().max()) / sc * 2,
          'own vs arena view', float((own[n] - net.arena.grad_views[n]).abs().max()) / sc, flush=True)
    dist.destroy_process_group()

if __name__ == '__main__':
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
