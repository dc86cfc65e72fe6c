"""dusty_gan_amd -- MI355X-native engine for the dusty-gan training hot path.

Mirrors the reference's user-facing surface for this path (`models.define_G/define_D`,
`trainers.dcgan_amp.Trainer`, `configs/`, `utils.diff_augment.DiffAugment`) on top of the C-ABI HIP library
`csrc/libdustygan_hip.so` (include/dusty_gan_hip.h).  There is no CPU fallback: forward/step need a GPU and the
built library.
"""
__version__ = "0.1.0"
