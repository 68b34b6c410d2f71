"""gazenerf_amd -- MI355X-native volumetric renderer for GazeNeRF's hot path.

Public surface:
    render_two_stream, render_two_stream_tiled,            (gazenerf_amd.render; the second: a training step over ray
    importance_resample, sample_zvals,                      tiles with the caller's per-ray loss, 3/3 of the FLOPs)
    PackedWeightCache
    merge_featmaps                                         (gazenerf_amd.merge)
    neural_render, NeuralRendererAMD                       (gazenerf_amd.upsample)
    HotPathRenderer, MLPParams, GazeNeRFNetAMD             (gazenerf_amd.module; the last one is the whole
                                                           reference network: hot path -> merge -> upsampler x4)
    losses                                                 image losses + per-sample fitting step (SURVEY 8(f) N4,
                                                           the terms without a pretrained network)
    gan                                                    PatchGAN discriminator, its loss / generator term / update step
                                                           (the GAN terms need no pretrained network)
    perceptual                                             VGG-perceptual terms: torchvision's vgg16.features layout + the reference's
                                                           loss arithmetic; ImageNet weights are the caller's
    data                                                   dataset-row contract, collate, prepare_batch: the reference's
                                                           sample -> op-input step
    synth                                                  synthetic input recipe
    build.build()                                          compile libgnr.so for gfx950
"""
from . import data, gan, losses, perceptual, synth  # noqa: F401
from .render import PackedWeightCache, importance_resample, render_two_stream, render_two_stream_tiled, sample_zvals  # noqa: F401
from .merge import merge_featmaps  # noqa: F401
from .upsample import NeuralRendererAMD, neural_render  # noqa: F401
from .module import GazeNeRFNetAMD, HotPathRenderer, MLPParams  # noqa: F401
