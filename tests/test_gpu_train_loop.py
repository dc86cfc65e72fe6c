"""The Trainer driven the way the reference's train.py drives it (train.py:59-166, minus Hydra / TensorBoard / wandb / the
BEV renderer, which are out of scope): construction from (cfg, local_cfg), the `real` preview through fetch_reals /
postprocess / A, the iteration loop over `trainer.step(i)` with the scalar dict, `generate()` at the image interval,
`validation()` at the test interval, `save_models()` at the checkpoint interval and at the end - then a second process-like
construction with `cfg.resume` that continues at `start_iteration + 1`.  Everything the loop touches is the drop-in
surface of SURVEY.md §8b."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_py_shaped_loop(tmp_path):
    from dusty_gan_amd.trainers import dcgan_amp
    from dusty_gan_amd.utils.config import load_config
    ngpus, gpu = 1, 0
    cfg = load_config(["model=dusty2_dcgan_eqlr", "dataset=synthetic", "dataset.shape=[32,64]", "model.gen.in_ch=16",
                       "model.gen.ch_base=8", "model.gen.ch_max=32", "model.dis.ch_base=8", "model.dis.ch_max=32",
                       "solver.batch_size=4", "solver.total_kimg=0.024", "solver.checkpoint.save_stats=1",
                       "solver.checkpoint.save_image=2", "solver.checkpoint.test=3", "solver.checkpoint.save_model=4",
                       "solver.validation.num_points=256", "enable_amp=false", "dataset.pool=2"])
    # train.py:52-66
    assert cfg.solver.batch_size % ngpus == 0
    local_batch_size = int(cfg.solver.batch_size / ngpus)
    assert local_batch_size % cfg.solver.num_accumulation == 0
    local_batch_size = int(local_batch_size / cfg.solver.num_accumulation)
    local_cfg = {"gpu": gpu, "ngpus": ngpus, "batch_size": local_batch_size,
                 "num_workers": int((cfg.num_workers + ngpus - 1) / ngpus)}
    torch.manual_seed(3)
    trainer = dcgan_amp.Trainer(cfg, local_cfg)                                  # :69
    total_img = cfg.solver.total_kimg * 1000
    total_iteration = int(total_img / cfg.solver.batch_size)                     # :71-73
    iteration_to_imgs = lambda i: int(i * cfg.solver.batch_size)
    assert total_iteration == 6 and trainer.start_iteration == 0
    H, W = cfg.dataset.shape

    # the `real` preview (:86-88)
    inv_real, mask_real = trainer.fetch_reals(next(trainer.loader))
    real = trainer.postprocess({"depth": inv_real, "mask": mask_real})
    real_aug = trainer.postprocess({"depth": trainer.A(inv_real)})
    assert real["depth"].shape == (local_batch_size, 1, H, W) and 0.0 <= float(real["depth"].min()) and float(real["depth"].max()) <= 1.0
    assert real["points"].shape == (local_batch_size, 3, H, W) and real["normals"].shape == (local_batch_size, 3, H, W)
    assert real_aug["depth"].shape == real["depth"].shape and str(trainer.device) == f"cuda:{gpu}"

    logged, images, scores, saved = [], [], [], []
    for i in range(trainer.start_iteration + 1, total_iteration + 1):           # :103-110
        scalars = trainer.step(i)
        step = iteration_to_imgs(i)
        if i % cfg.solver.checkpoint.save_stats == 0:                            # :117-120
            for key, scalar in scalars.items():
                assert isinstance(key, str) and isinstance(scalar, float) and math.isfinite(scalar), (key, scalar)
            logged.append(dict(scalars.items()))
        if i % cfg.solver.checkpoint.save_image == 0:                            # :123-151
            out = trainer.generate()
            assert {"depth", "points", "normals", "depth_orig", "confidence", "mask"} <= set(out)
            assert out["confidence"].shape[1] == 2 and out["mask"].shape[1] == 2
            images.append(step)
        if i % cfg.solver.checkpoint.test == 0:                                  # :154-157
            sc = trainer.validation()
            assert all(isinstance(v, float) for v in sc.values())
            assert {"jsd", "mmd-cd", "cov-cd", "1-nn-accuracy-cd"} <= set(sc) and any(k.startswith("swd") for k in sc)
            scores.append(sc)
        if i % cfg.solver.checkpoint.save_model == 0:                            # :160-161
            saved.append(trainer.save_models("{:010d}".format(int(step)), int(step), directory=str(tmp_path)))
    step = iteration_to_imgs(total_iteration)                                    # :164-166
    final = trainer.save_models("{:010d}".format(int(step)), int(step), directory=str(tmp_path))
    assert len(logged) == 6 and images == [8, 16, 24] and len(scores) == 2 and len(saved) == 1
    assert set(logged[0]) == {"loss/D/output/real", "loss/D/output/fake", "loss/D/adversarial", "loss/D/gradient_penalty",
                              "loss/G/adversarial"}                              # the reference's scalar keys (:319-323)
    assert os.path.basename(final) == "checkpoint_0000000024.pth"

    # a later invocation with cfg.resume (:134-144): continues behind the last finished iteration
    cfg2 = load_config(["model=dusty2_dcgan_eqlr", "dataset=synthetic", "dataset.shape=[32,64]", "model.gen.in_ch=16",
                        "model.gen.ch_base=8", "model.gen.ch_max=32", "model.dis.ch_base=8", "model.dis.ch_max=32",
                        "solver.batch_size=4", "solver.total_kimg=0.032", "enable_amp=false", "dataset.pool=2"])
    cfg2.resume = final
    t2 = dcgan_amp.Trainer(cfg2, local_cfg)
    assert t2.start_iteration == 6
    for net in ("G", "D", "G_ema"):
        assert torch.equal(getattr(t2, net).store.flat.cpu(), getattr(trainer, net).store.flat.cpu()), net
    total2 = int(cfg2.solver.total_kimg * 1000 / cfg2.solver.batch_size)
    ran = [dict(t2.step(i).items()) for i in range(t2.start_iteration + 1, total2 + 1)]
    assert len(ran) == 2 and all(math.isfinite(v) for s in ran for v in s.values())
    assert t2.optim_G.step_count == 8
