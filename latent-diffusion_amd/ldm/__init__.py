"""MI355X-native drop-in for the hot path of anenbergb/latent-diffusion.

Put ``latent-diffusion_amd/`` on ``sys.path`` and the trainer's own imports
(``from ldm.unet import UNet``, ``from ldm.scheduling_ddpm import DDPMScheduler``,
train_latent_diffusion.py:37,:41) resolve to the HIP-backed implementations.
"""
