"""The reference's test_svi.py (image-to-video rolling window, /root/reference/test_svi.py:65-485) on the HIP backend, without diffsynth's
ModelManager: checkpoints go straight into HBM (svi_hip.checkpoint), the SVI LoRA is merged on the device (svi_hip.lora), prompts and the
reference image are encoded by the HIP umT5 / CLIP encoders, and the clip loop runs resident on the GPU (svi_hip.StreamLoop: conditioning
encode -> 50-step CFG denoise on one captured step graph -> decode -> 8-bit frames -> motion-frame hand-off -> stitching).

The argument surface is the reference script's (same names, defaults and meaning; --num_persistent_param_in_dit is accepted and ignored:
nothing is offloaded on a 288 GB part).  Added:
    --synthetic                 no checkpoints, no tokenizer, no image files: random-init weights of the named architecture, seeded random prompt
                                embeddings, a synthetic reference image — the whole chain runs on a box that has neither weights nor network
    --synthetic_model NAME      tiny-i2v (seconds; the test suite's toy widths), 14b-i2v (Wan2.1-I2V-14B, the model test_svi.py runs; default)
    --height / --width          synthetic image size (default 480 x 832; the reference derives it from the image file, utils/image_process.py:39-70)
Outputs: <output>/<name>_<timestamp>/video_u8.npy ([frames, H, W, 3] uint8, the stitched window) and first / last frame as PNG;
the reference writes an mp4 through imageio, which this image does not have.

    python examples/test_svi_hip.py --synthetic --synthetic_model tiny-i2v --num_clips 3 --num_steps 4
    python examples/test_svi_hip.py --dit_root weights/Wan2.1-I2V-14B-480P/ --extra_module_root weights/Stable-Video-Infinity/version-1.0/svi-shot.safetensors \\
        --ref_image_path data/cat.png --prompt_path data/cat_prompt.txt --num_clips 10

To keep the reference's own script instead, add ONE line to it — `import svi_hip; svi_hip.install(pipe)` after `from_model_manager(...)`
(before or after its `pipe.enable_vram_management(...)`, which then does nothing) — see INTEGRATION.md.
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import sys
import time
from datetime import datetime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "stable-video-infinity_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

COMMON_NEGATIVE_PROMPT = ("bright tones, overexposed, static, blurred details, subtitles, style, works, paintings, images, static, overall gray, worst quality, "
                          "low quality, JPEG compression residue, ugly, incomplete, extra fingers, poorly drawn hands, poorly drawn faces, deformed, disfigured, "
                          "misshapen limbs, fused fingers, still picture, messy background, three legs, many people in the background, walking backwards")


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="SVI image-to-video rolling window on the HIP backend (argument surface of the reference's test_svi.py).")
    ap.add_argument("--dit_root", default="weights/Wan2.1-I2V-14B-480P/", type=str, help="Root directory of the Wan2.1-I2V model.")
    ap.add_argument("--extra_module_root", default="weights/Stable-Video-Infinity/version-1.0/svi-shot.safetensors", type=str)
    ap.add_argument("--output", default="videos/", type=str)
    ap.add_argument("--cfg_scale_text", default=5.0, type=float)
    ap.add_argument("--lora_alpha", default=1.0, type=float)
    ap.add_argument("--train_architecture", default="lora", type=str)
    ap.add_argument("--ref_pad_cfg", default=False, action="store_true", help="Whether to set mask with only 1-frame 1.")
    ap.add_argument("--num_motion_frames", type=int, default=1)
    ap.add_argument("--num_clips", type=int, default=10)
    ap.add_argument("--num_steps", type=int, default=50)
    ap.add_argument("--data_root", type=str, default="data_inference/wan_i2v/")
    ap.add_argument("--ref_image_path", type=str, default=None)
    ap.add_argument("--prompt_path", type=str, default=None)
    ap.add_argument("--test_samples", type=str, nargs="*")
    ap.add_argument("--max_prompts_per_sample", type=int, default=None)
    ap.add_argument("--ref_pad_num", type=int, default=0, help="0 -> no padding, k -> padding k, -1 -> full padding")
    ap.add_argument("--use_first_prompt_only", default=False, action="store_true")
    ap.add_argument("--use_first_aug", default=False, action="store_true")
    ap.add_argument("--max_width", type=int, default=832)
    ap.add_argument("--seed_times", type=int, default=42)
    ap.add_argument("--repeat_first_clip", default=False, action="store_true")
    ap.add_argument("--tiled", default=False, action="store_true")
    ap.add_argument("--tile_size", type=int, nargs="+", default=[30, 52])
    ap.add_argument("--tile_stride", type=int, nargs="+", default=[15, 26])
    ap.add_argument("--prompt_prefix", type=str, default="none")
    ap.add_argument("--prompt_repeat_times", type=int, default=1)
    ap.add_argument("--num_persistent_param_in_dit", type=int, default=6 * 10 ** 9, help="accepted for compatibility; nothing is offloaded (288 GB HBM)")
    ap.add_argument("--synthetic", action="store_true", help="random-init weights, random prompt embeddings, a synthetic image: no files needed")
    ap.add_argument("--synthetic_model", default="14b-i2v", choices=["tiny-i2v", "14b-i2v"])
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--max_frames", type=int, default=81, help="frames per clip (the reference's module constant max_frames = 81)")
    return ap.parse_args(argv)


def load_prompts_from_file(path: str):
    """`prompts = [...]` assignment or one prompt per line (test_svi.py:30-62)."""
    default = ["Default prompt: the subject is moving naturally"]
    if not os.path.exists(path):
        print(f"Warning: prompt file not found at {path}")
        return default
    text = open(path, "r", encoding="utf-8").read()
    at = text.find("prompts = [")
    if at >= 0:
        import ast
        try:
            return list(ast.literal_eval(text[text.index("[", at):text.rindex("]") + 1])) or default
        except (ValueError, SyntaxError):
            pass
    lines = [ln.strip() for ln in text.split("\n") if ln.strip() and not ln.strip().startswith("#")]
    return lines or default


def calculate_dimensions(width: int, height: int, max_width: int):
    """utils/image_process.py:39-70 on a size: fit under max_width keeping the aspect ratio, both rounded down to multiples of 16."""
    if width > max_width:
        height = int(max_width * (height / width))
        width = max_width
    return (height // 16) * 16, (width // 16) * 16


# ------------------------------------------------------------------------------------------------------------------ synthetic models / inputs
def synthetic_models(name: str, dev):
    """(dit, vae, clip_encoder, embed(prompt)) with random-init weights of the named architecture."""
    import synth
    import svi_hip
    from svi_hip.vae import WanVideoVAE, device_vae_weights
    cfg = dict(synth.TINY_DIT_I2V if name == "tiny-i2v" else synth.WAN_14B_I2V)
    sys.path.insert(0, ROOT)
    from bench import device_weights
    dit = svi_hip.WanDiT(eps=1e-6, num_heads=synth.num_heads_of(cfg), **cfg)
    dit.bind(device_weights(cfg, 0, dev))
    vae = WanVideoVAE.from_state_dict(device_vae_weights(0, dev))
    text_dim = cfg["text_dim"]

    def embed(prompt: str) -> torch.Tensor:
        """A seeded stand-in for umT5(prompt): as many non-zero rows as the prompt has words (capped), zero padding behind — the shape and the
        zero tail the prompter produces (prompters/wan_prompter.py:101-112)."""
        seed = int.from_bytes(hashlib.sha256(prompt.encode()).digest()[:4], "little")
        g = torch.Generator(device=dev).manual_seed(seed)
        e = torch.randn((1, 512, text_dim), generator=g, device=dev)
        e[:, min(511, max(4, len(prompt.split()))):] = 0
        return e.to(torch.bfloat16)

    def clip_encoder(first: torch.Tensor) -> torch.Tensor:          # stand-in for CLIP(first frame): depends on the frame, [1, 257, 1280]
        base = torch.randn((1, 257, 1280), generator=torch.Generator(device=dev).manual_seed(11), device=dev)
        return (base + first.float().mean()).to(torch.bfloat16)
    return dit, vae, clip_encoder, embed


def real_models(args, dev):
    """Checkpoints -> HBM -> HIP handles; the SVI LoRA merged on the device; umT5 / CLIP on the HIP encoders."""
    import svi_hip
    from svi_hip import checkpoint, lora
    root = args.dit_root
    shards = sorted(glob.glob(os.path.join(root, "diffusion_pytorch_model-*.safetensors"))) or sorted(glob.glob(os.path.join(root, "*.safetensors")))
    if not shards:
        raise SystemExit(f"no DiT shards under {root} (or run with --synthetic)")
    dit = checkpoint.load_dit(shards, device=dev)
    files = [args.extra_module_root] if args.extra_module_root.endswith(".safetensors") else sorted(glob.glob(os.path.join(args.extra_module_root, "*.safetensors")))
    for f in files:
        n = lora.load_lora_(dit, checkpoint.load_safetensors(f, device=dev), alpha=args.lora_alpha)
        print(f"    {n} tensors are updated by {os.path.basename(f)}.")
    vae = checkpoint.load_vae(os.path.join(root, "Wan2.1_VAE.pth"), device=dev)
    text = checkpoint.load_text_encoder(os.path.join(root, "models_t5_umt5-xxl-enc-bf16.pth"), device=dev)
    clip = checkpoint.load_image_encoder(os.path.join(root, "models_clip_open-clip-xlm-roberta-large-vit-huge-14.pth"), device=dev)
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(os.path.join(root, "google/umt5-xxl"))

    def embed(prompt: str) -> torch.Tensor:
        enc = tok([" ".join(prompt.split())], padding="max_length", truncation=True, max_length=512, add_special_tokens=True, return_tensors="pt")
        return text.forward(enc.input_ids, enc.attention_mask, rows="valid")        # padded rows come back zero (prompter:110-111)
    return dit, vae, clip, embed


# ------------------------------------------------------------------------------------------------------------------ scenarios
def scenarios(args):
    if args.synthetic:
        prompts = [f"synthetic prompt {i}: the subject keeps moving through scene number {i}" + " and on" * i for i in range(4)]
        return [dict(name="synthetic", image=None, prompts=prompts)]
    if args.ref_image_path and args.prompt_path:
        name = os.path.splitext(os.path.basename(args.ref_image_path))[0]
        return [dict(name=name, image=args.ref_image_path, prompts=load_prompts_from_file(args.prompt_path))]
    out = []
    for d in sorted(os.listdir(args.data_root)):
        full = os.path.join(args.data_root, d)
        if not os.path.isdir(full) or (args.test_samples and d not in args.test_samples):
            continue
        imgs = [f for f in sorted(os.listdir(full)) if f.lower().endswith((".png", ".jpg", ".jpeg", ".webp"))]
        if imgs:
            out.append(dict(name=d, image=os.path.join(full, imgs[0]), prompts=load_prompts_from_file(os.path.join(full, "prompt.txt"))))
    if not out:
        raise SystemExit(f"no test samples under {args.data_root}")
    return out


def main(argv=None) -> dict:
    args = parse_args(argv)
    import svi_hip
    assert torch.cuda.is_available(), "the HIP backend needs a GPU"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    t0 = time.perf_counter()
    dit, vae, clip_encoder, embed = synthetic_models(args.synthetic_model, dev) if args.synthetic else real_models(args, dev)
    torch.cuda.synchronize()
    print(f"models resident after {time.perf_counter() - t0:.1f} s ({'synthetic ' + args.synthetic_model if args.synthetic else args.dit_root})")
    os.makedirs(args.output, exist_ok=True)
    summary = []
    for sc in scenarios(args):
        prompts = sc["prompts"][:args.max_prompts_per_sample] if args.max_prompts_per_sample else sc["prompts"]
        if args.prompt_prefix != "none":
            prompts = [f"{args.prompt_prefix}, {p}" for p in prompts]
        if sc["image"] is None:
            height, width = calculate_dimensions(args.width, args.height, args.max_width)
            rs = np.random.RandomState(0)
            yy, xx = np.mgrid[0:height, 0:width]
            img = np.stack([(xx * 255 // max(width - 1, 1)), (yy * 255 // max(height - 1, 1)), rs.randint(0, 256, (height, width))], axis=-1).astype(np.uint8)
        else:
            from PIL import Image
            pil = Image.open(sc["image"]).convert("RGB")
            height, width = calculate_dimensions(pil.size[0], pil.size[1], args.max_width)
            img = np.asarray(pil.resize((width, height)), dtype=np.uint8)
        # number of clips: the reference's rule (test_svi.py:394-399)
        num_clips = args.num_clips if args.use_first_prompt_only else min(args.num_clips, len(prompts) * args.prompt_repeat_times)
        neg = embed(COMMON_NEGATIVE_PROMPT)
        embedded = [(embed(p), neg) for p in (prompts[:1] if args.use_first_prompt_only else prompts)]
        frame = torch.from_numpy(img)
        first = frame[None].repeat(args.num_motion_frames, 1, 1, 1) if args.repeat_first_clip else frame[None]
        loop = svi_hip.StreamLoop(dit, vae, clip_encoder=clip_encoder, num_motion_frames=args.num_motion_frames, num_frames=args.max_frames,
                                  num_inference_steps=args.num_steps, cfg_scale=args.cfg_scale_text, ref_pad_cfg=args.ref_pad_cfg, ref_pad_num=args.ref_pad_num,
                                  seed_times=args.seed_times, tiled=args.tiled, tile_size=tuple(args.tile_size), tile_stride=tuple(args.tile_stride))
        print(f"\n{'#' * 80}\nSAMPLE {sc['name']}: {width}x{height}, {num_clips} clips x {args.max_frames} frames, {args.num_steps} steps, {len(embedded)} prompt(s)")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        video = loop.run(first, frame, embedded, num_clips, prompt_repeat_times=args.prompt_repeat_times, use_first_prompt_only=args.use_first_prompt_only)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        latent_frames = num_clips * ((args.max_frames - 1) // 4 + 1)
        out_dir = os.path.join(args.output, f"{sc['name']}_{datetime.now().strftime('%Y%m%d_%H%M%S')}")
        os.makedirs(out_dir, exist_ok=True)
        vid = video.cpu().numpy()
        np.save(os.path.join(out_dir, "video_u8.npy"), vid)
        try:
            from PIL import Image
            Image.fromarray(vid[0]).save(os.path.join(out_dir, "frame_first.png"))
            Image.fromarray(vid[-1]).save(os.path.join(out_dir, "frame_last.png"))
        except Exception as ex:      # PNGs are a convenience
            print(f"(no PNG written: {type(ex).__name__})")
        rec = dict(sample=sc["name"], clips=num_clips, frames=int(vid.shape[0]), height=height, width=width, seconds=round(dt, 3),
                   latent_frames_per_s=round(latent_frames / dt, 4), step_graph_captures=loop.loop.captures, seeds=[t["seed"] for t in loop.trace], out=out_dir)
        print(f"window done in {dt:.1f} s = {rec['latent_frames_per_s']} latent frames/s; {vid.shape[0]} frames -> {out_dir}; step graph captured {loop.loop.captures} time(s)")
        summary.append(rec)
    print(json.dumps({"test_svi_hip": summary}))
    return {"test_svi_hip": summary}


if __name__ == "__main__":
    main()
