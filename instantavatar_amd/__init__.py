"""instantavatar_amd -- MI355X-native hot path of InstantAvatar.

Plugin classes mirror the reference's `_target_` surface
(confs/deformer/fast_snarf.yaml, confs/network/ngp.yaml,
confs/renderer/raymarcher_acc.yaml):

    instantavatar_amd.deformers.snarf_deformer.SNARFDeformer
    instantavatar_amd.models.networks.ngp.NeRFNGPNet
    instantavatar_amd.renderers.raymarcher_acc.Raymarcher

Compute goes through libinstantavatar_hip.so (C ABI: include/instantavatar_hip.h).
"""
__version__ = "0.1.0"
