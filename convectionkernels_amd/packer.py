"""Image -> compressed texture file on the MI355X: the caller side of the hot path.

    python -m convectionkernels_amd.packer [-format F] [-uniform] [-fakebt709] [-quality Q] [-dds] input output

The command line follows the reference's example packer (etc2packer.cpp:44-105: `-format etc1|etc2rgb|etc2rgba|etc2punchthrough|r11u|r11s`,
`-fakebt709`, `-uniform` (which overrides it), input, output; default etc2rgb, KTX output) and adds the BC formats (bc1..bc5, bc7; `-dds` for a DX10 DDS
file, `-quality 1..100` for a BC7 plan).  The image is uploaded once; tiling into groups of eight 4x4 blocks with
edge clamping (etc2packer.cpp:215-248), encoding and the removal of padding blocks all run on the device, and the
packed blocks are already in container order.  The input is anything PIL opens, or a .npy of shape (H, W, 4) uint8."""
import sys

import numpy as np

from . import api, container

USAGE = __doc__.split("\n\n")[1]


def load_rgba8(path):
    if path.endswith(".npy"):
        img = np.load(path)
    else:
        from PIL import Image  # only needed for image files
        img = np.array(Image.open(path).convert("RGBA"))
    if img.ndim != 3 or img.shape[2] != 4 or img.dtype != np.uint8:
        raise ValueError("expected an (H, W, 4) uint8 image, got %s %s" % (img.shape, img.dtype))
    return np.ascontiguousarray(img)


def r11_blocks(img, signed):
    """PixelBlockScalarS16 tiles of the RGB average exactly as etc2packer.cpp:236-241 computes them (including its
    use of the *unsigned* normalised value, scaled by 1023, for the signed format), groups of 8 blocks, edges clamped."""
    h, w = img.shape[:2]
    bw, bh = (w + 3) // 4, (h + 3) // 4
    gw = (bw + 7) // 8 * 8
    ys = np.minimum(np.arange(bh * 4), h - 1)
    xs = np.minimum(np.arange(gw * 4), w - 1)
    total = img[ys][:, xs, :3].astype(np.float64).sum(axis=2)
    normalized = total / (255.0 * 3.0)
    values = np.floor(normalized * (1023.0 if signed else 2047.0) + 0.5).astype(np.int16)
    return values.reshape(bh, 4, gw, 4).transpose(0, 2, 1, 3).reshape(bh * gw, 16), bw, bh, gw


def encode_file(image, fmt, options=None, plan=None, ctx=None):
    """(H, W, 4) uint8 numpy image -> packed blocks (ceil(H/4) * ceil(W/4), bytesPerBlock) uint8, container order."""
    import torch
    ctx = ctx or api.default_context()
    fmt = container.canonical(fmt)
    h, w = image.shape[:2]
    if fmt in ("r11u", "r11s"):
        blocks, bw, bh, gw = r11_blocks(image, fmt == "r11s")
        packed = ctx.encode_etc2_alpha11(blocks, signed=(fmt == "r11s"), options=options)
        return np.asarray(packed).reshape(bh, gw, 8)[:, :bw].reshape(-1, 8)
    dev = torch.from_numpy(image).cuda(ctx.device)
    packed = ctx.encode_image(fmt, dev, options, plan)
    torch.cuda.synchronize(ctx.device)
    return packed.cpu().numpy()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    fmt, uniform, fake, quality, dds, paths = "etc2rgb", False, False, None, False, []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "-format" and i + 1 < len(argv):
            fmt = argv[i + 1]
            i += 1
        elif a == "-quality" and i + 1 < len(argv):
            quality = int(argv[i + 1])
            i += 1
        elif a == "-uniform":
            uniform = True
        elif a == "-dds":
            dds = True
        elif a == "-fakebt709":
            fake = True
        elif a.startswith("-"):
            sys.stderr.write(USAGE + "\n")
            return 2
        else:
            paths.append(a)
        i += 1
    if len(paths) != 2:
        sys.stderr.write(USAGE + "\n")
        return 2
    try:
        fmt = container.canonical(fmt)
        image = load_rgba8(paths[0])
    except (ValueError, OSError) as e:
        sys.stderr.write("%s\n" % e)
        return 1
    options = api.Options()
    if uniform:  # etc2packer.cpp:202-205
        options.flags |= api.Flags.Uniform
    elif fake:
        options.flags |= api.Flags.ETC_UseFakeBT709
    plan = None
    if quality is not None:
        plan = api.BC7EncodingPlan()
        api.ConfigureBC7EncodingPlanFromQuality(plan, quality)
    packed = encode_file(image, fmt, options, plan)
    h, w = image.shape[:2]
    (container.write_dds if dds else container.write_ktx)(paths[1], fmt, w, h, packed)
    return 0


if __name__ == "__main__":
    sys.exit(main())
