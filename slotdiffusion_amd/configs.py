"""Configuration values of the reference's shipped `*_params.py` files for the hot-path models, restated
as plain dicts (values only; the reference files are classes over nerv's BaseParams and also run
unchanged through `slotdiffusion_amd.compat.install_nerv_shim()`).  Used by bench.py, the tests and
`__graft_entry__.smoke()`; every dict is checked against the dump of the reference's own file
(`tools/dump_ref_configs.py` -> tests/golden/configs/*.json) in tests/test_configs_cpu.py."""
import copy


def clevrtex_cfg(num_slots=7):
    """Restates img_based/configs/sa_ldm/sa_ldm_clevrtex_params-res128.py (values only)."""
    res = (128, 128)
    d = 192
    return dict(
        resolution=res,
        slot_dict=dict(num_slots=num_slots, slot_size=d, slot_mlp_size=2 * d, num_iterations=3),
        enc_dict=dict(resnet='resnet18', use_layer4=False, enc_out_channels=d),
        dec_dict=dict(
            resolution=(32, 32),
            vae_dict=dict(
                vae_type='VQVAE',
                enc_dec_dict=dict(resolution=128, in_channels=3, z_channels=3, ch=64,
                                  ch_mult=[1, 2, 4], num_res_blocks=2, attn_resolutions=[],
                                  out_ch=3, dropout=0.0),
                vq_dict=dict(n_embed=4096, embed_dim=3, percept_loss_w=1.0),
                vqvae_ckp_path='./pretrained/vqvae_clevrtex_params-res128.pth'),
            unet_dict=dict(in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2,
                           attention_resolutions=(8, 4, 2), dropout=0.1,
                           channel_mult=(1, 2, 3, 4), dims=2, use_checkpoint=False,
                           num_head_channels=32, resblock_updown=False, conv_resample=True,
                           transformer_depth=1, context_dim=d, n_embed=None),
            use_ema=False,
            diffusion_dict=dict(pred_target='eps', z_scale_factor=1., timesteps=1000,
                                beta_schedule='linear', linear_start=0.0015, linear_end=0.0195,
                                cosine_s=8e-3, log_every_t=200, logvar_init=0.),
            conditioning_key='crossattn', cond_stage_key='slots'),
        loss_dict=dict(use_denoise_loss=True))


def sa_plain_cfg(num_slots=7):
    """Restates img_based/configs/sa/sa_clevrtex_params-res128.py (values only; BASELINE.json
    config 0 asks 7 slots)."""
    d = 192
    return dict(resolution=(128, 128),
                slot_dict=dict(num_slots=num_slots, slot_size=d, slot_mlp_size=2 * d, num_iterations=3),
                enc_dict=dict(resnet='resnet18', use_layer4=False, enc_out_channels=d),
                dec_dict=dict(dec_channels=(d, 128, 128, 128, 128), dec_resolution=(8, 8), dec_ks=5,
                              dec_norm=''),
                loss_dict=dict(use_img_recon_loss=True))


def savi_cfg():
    """Restates video_based/configs/savi/savi_movie_params-res128.py (values only)."""
    d = 192
    return dict(resolution=(128, 128), clip_len=3,
                slot_dict=dict(num_slots=15, slot_size=d, slot_mlp_size=2 * d, num_iterations=2),
                enc_dict=dict(resnet='resnet18', use_layer4=False, enc_out_channels=d,
                              replace_stride_with_dilation=[False, False, False]),
                dec_dict=dict(dec_channels=(d, 64, 64, 64, 64), dec_resolution=(8, 8), dec_ks=5,
                              dec_norm=''),
                pred_dict=dict(pred_type='transformer', pred_rnn=False, pred_norm_first=True,
                               pred_num_layers=2, pred_num_heads=4, pred_ffn_dim=4 * d,
                               pred_sg_every=None),
                loss_dict=dict(use_img_recon_loss=True))


def movie_cfg():
    """Restates video_based/configs/savi_ldm/savi_ldm_movie_params-res128.py (values only)."""
    cfg = clevrtex_cfg(num_slots=15)
    cfg['slot_dict']['num_iterations'] = 2
    cfg['clip_len'] = 3
    cfg['pred_dict'] = dict(pred_type='transformer', pred_rnn=False, pred_norm_first=True,
                            pred_num_layers=2, pred_num_heads=4, pred_ffn_dim=768,
                            pred_sg_every=None)
    cfg['dec_dict']['vae_dict']['vqvae_ckp_path'] = './pretrained/vqvae_movie_params-res128.pth'
    return cfg


def movid_cfg(num_slots=11, clip_len=6):
    """video_based/configs/savi_ldm/savi_ldm_movid_params-res128.py with the two values BASELINE.json
    config 2 overrides (11 slots, 6-frame clips; the file ships 15 / 3).  Everything else equals
    the MOVi-E config (only dataset level and checkpoint path differ)."""
    cfg = movie_cfg()
    cfg['slot_dict']['num_slots'] = num_slots
    cfg['clip_len'] = clip_len
    cfg['dec_dict']['vae_dict']['vqvae_ckp_path'] = './pretrained/vqvae_movid_params-res128.pth'
    return cfg


def dino_coco_cfg():
    """img_based/configs/sa_ldm/sa_ldm_dino_coco_params-res224.py (BASELINE config 5): 224 x 224,
    frozen DINO ViT-S/8 features 28 x 28 x 384, 7 slots of 256, latent 56 x 56, UNet context 256."""
    cfg = clevrtex_cfg(num_slots=7)
    cfg['resolution'] = (224, 224)
    cfg['slot_dict'] = dict(num_slots=7, slot_size=256, slot_mlp_size=512, num_iterations=3)
    cfg['enc_dict'] = dict(dino='dino-vits8', patch_size=8, small_size=True, resolution=(224, 224),
                           enc_out_channels=256)
    dd = cfg['dec_dict']
    dd['resolution'] = (56, 56)
    dd['vae_dict']['enc_dec_dict']['resolution'] = 224
    dd['vae_dict']['vqvae_ckp_path'] = './pretrained/vqvae_coco_params-res224.pth'
    dd['unet_dict']['context_dim'] = 256
    return cfg


# bench.py --config names -> (builder, frames per clip or None, default batch per GPU, clip_grad)
BENCH_CONFIGS = {
    'clevrtex128': dict(cfg=lambda: clevrtex_cfg(num_slots=7), frames=None, batch=64, clip_grad=1.0,
                        baseline='configs[1]: img_based SlotDiffusion, CLEVRTex 128x128, 7 slots'),
    'coco224': dict(cfg=dino_coco_cfg, frames=None, batch=16, clip_grad=0.05,
                    baseline='configs[4]: img_based DINOSAUR-init SlotDiffusion, COCO 224x224, 7 slots'),
    'movid11x6': dict(cfg=lambda: movid_cfg(num_slots=11, clip_len=6), frames=6, batch=16, clip_grad=0.05,
                      baseline='configs[2]: video_based SAVi+LDM, MOVi-D 128x128x6 frames, 11 slots'),
    'movie15x6': dict(cfg=lambda: dict(movie_cfg(), clip_len=6), frames=6, batch=16, clip_grad=0.05,
                      baseline='configs[3]: video_based SlotDiffusion, MOVi-E 128x128x6 frames, 15 slots'),
}
