from . import AP2POH, RGBD2AP, data_loader, discriminator, generator, loss_func, perceptual, watermelon  # noqa: F401
