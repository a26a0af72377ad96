### every launch line of the reference's `command` file, for one 8xMI355X node
### (torchrun == what torchx `dist.ddp -j NxM` runs per replica; DDL_* env vars relocate /mnt/dcornelius)

### single
python single.py
python -m torchx_component.submit_single python --script single.py

### ddp   (reference: -j 1x2, 2x1, 3x1, 3x2)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ddp.py
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 ddp.py
python -m torchx_component.submit_single ddp --script ddp.py -j 1x8

### pp    (2-stage, and the 4-stage block-BEGINNING split of debug.py)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 pp.py
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 pp.py --split-points features.denseblock2.denselayer1 features.denseblock3.denselayer1 features.denseblock4.denselayer1

### ddp n pp   (reference: 3x2 mesh on 6 GPUs; 4x2 on one MI355X node)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 ddp_n_pp.py --pp-size 2
python -m torch.distributed.run --nnodes=1 --nproc-per-node 6 --master-addr 127.0.0.1 ddp_n_pp.py --pp-size 2

### communication time  (reference ping + message-size sweep)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 communication_time.py --sweep

### single with ddp
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 ddp.py

### running test.py (mesh smoke) / debug.py (split explorer)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 6 --master-addr 127.0.0.1 test.py
python debug.py

### headline benchmark
python bench.py --gpus 1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 bench.py --gpus 8

### round 2: balanced pipelines and throughput-sized shapes
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 pp.py --split-points auto --schedule 1f1b
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 ddp_n_pp.py --pp-size 2 --split-points auto --batch-size 240 --num-microbatches 8
python debug.py --auto-stages 2 4 --micro-batch 30
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 bench.py --gpus 2 --mode pp --preset mi355x
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 bench.py --gpus 8 --mode ddpnpp --pp-size 2 --preset mi355x-large
python bench.py --batch-per-gpu 1024        # tensors beyond 2 GiB
