# ROCm 7 + PyTorch-ROCm image for MI355X (gfx950); the reference used pytorch:2.6.0-cuda11.8 (Dockerfile:1)
FROM rocm/pytorch:rocm7.0_ubuntu22.04_py3.10_pytorch_2.10
WORKDIR /workspace
COPY . /workspace
RUN pip install -r requirements.txt && PYTORCH_ROCM_ARCH=gfx950 python __graft_entry__.py
ENV HSA_ENABLE_IPC_MODE_LEGACY=0
