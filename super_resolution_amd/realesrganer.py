"""RealESRGANer -- drop-in for Real-ESRGAN-master/realesrgan/utils.py:14-263 of the reference.

Same constructor, attributes (`scale, tile_size, tile_pad, pre_pad, mod_scale, half, device, model,
img, output, mod_pad_h, mod_pad_w`) and methods (`dni, pre_process, process, tile_process,
post_process, enhance`).  Differences, all behind the same interface:
  * `half=True` selects bf16 MFMA compute inside the network (parameters stay float32); the image
    tensors handed to the model are float32 either way;
  * cv2 is not required: the colour conversions of `enhance` are numpy (COLOR_BGR2GRAY = 0.114 B +
    0.587 G + 0.299 R); `outscale != scale` (cv2 LANCZOS4 resize, utils.py:256-261) raises;
  * when torch.distributed is initialised with world_size > 1 and `shard_tiles=True`, tile_process
    splits the tiles over the ranks (largest first, round-robin) and rank 0 gathers the pasted
    centres with point-to-point transfers -- tiles are independent, so there is no collective;
  * a failed tile raises instead of pasting a stale tile (utils.py:158-162 latent bug).
"""
import math

import numpy as np
import torch
from torch.nn import functional as F


class RealESRGANer():

    def __init__(self, scale, model_path, dni_weight=None, model=None, tile=0, tile_pad=10, pre_pad=10, half=False,
                 device=None, gpu_id=None, shard_tiles=False, tile_batch=8):
        """The reference signature (utils.py:29-40) plus two keyword extensions: `shard_tiles` deals the tiles over the
        ranks of an initialised process group, `tile_batch` runs up to that many tiles of equal window size through the
        network in one batch (every statistic of the network is per sample, so the result is the one-by-one result)."""
        self.scale = scale
        self.tile_batch = max(1, int(tile_batch))
        self.tile_size = tile
        self.tile_pad = tile_pad
        self.pre_pad = pre_pad
        self.mod_scale = None
        self.half = half
        self.shard_tiles = shard_tiles
        if gpu_id:
            self.device = torch.device(f'cuda:{gpu_id}' if torch.cuda.is_available() else 'cpu') if device is None else device
        else:
            self.device = torch.device('cuda' if torch.cuda.is_available() else 'cpu') if device is None else device
        if model is None:
            raise ValueError('RealESRGANer needs the network (model=...)')
        if model_path is not None:
            if isinstance(model_path, list):
                assert len(model_path) == len(dni_weight), 'model_path and dni_weight should have the save length.'
                loadnet = self.dni(model_path[0], model_path[1], dni_weight)
            else:
                if model_path.startswith('https://'):
                    raise RuntimeError('no network access: download %s yourself and pass the local path' % model_path)
                loadnet = torch.load(model_path, map_location=torch.device('cpu'), weights_only=True)
            keyname = 'params_ema' if 'params_ema' in loadnet else 'params'
            model.load_state_dict(loadnet[keyname], strict=True)
        model.eval()
        self.model = model.to(self.device)
        if self.half:
            if hasattr(self.model, 'compute_dtype'):
                self.model.compute_dtype = torch.bfloat16
            else:
                self.model = self.model.half()

    def dni(self, net_a, net_b, dni_weight, key='params', loc='cpu'):
        """Deep network interpolation of two checkpoints (utils.py:77-86)."""
        net_a = torch.load(net_a, map_location=torch.device(loc), weights_only=True)
        net_b = torch.load(net_b, map_location=torch.device(loc), weights_only=True)
        for k, v_a in net_a[key].items():
            net_a[key][k] = dni_weight[0] * v_a + dni_weight[1] * net_b[key][k]
        return net_a

    def _model_dtype(self):
        return torch.float16 if (self.half and not hasattr(self.model, 'compute_dtype')) else torch.float32

    def pre_process(self, img):
        """HWC float numpy -> 1CHW tensor on the device, reflect pre-pad and mod-pad (utils.py:88-111)."""
        img = torch.from_numpy(np.ascontiguousarray(np.transpose(img, (2, 0, 1)))).float()
        self.img = img.unsqueeze(0).to(self.device).to(self._model_dtype())
        if self.pre_pad != 0:
            self.img = F.pad(self.img, (0, self.pre_pad, 0, self.pre_pad), 'reflect')
        if self.scale == 2:
            self.mod_scale = 2
        elif self.scale == 1:
            self.mod_scale = 4
        if self.mod_scale is not None:
            self.mod_pad_h, self.mod_pad_w = 0, 0
            _, _, h, w = self.img.size()
            if h % self.mod_scale != 0:
                self.mod_pad_h = self.mod_scale - h % self.mod_scale
            if w % self.mod_scale != 0:
                self.mod_pad_w = self.mod_scale - w % self.mod_scale
            self.img = F.pad(self.img, (0, self.mod_pad_w, 0, self.mod_pad_h), 'reflect')

    def process(self):
        self.output = self.model(self.img)

    def tile_windows(self):
        """(input window, output core, offset of the core inside the window) per tile, in the reference's loop order."""
        _, _, height, width = self.img.shape
        wins = []
        for y in range(math.ceil(height / self.tile_size)):
            for x in range(math.ceil(width / self.tile_size)):
                x0, y0 = x * self.tile_size, y * self.tile_size
                x1, y1 = min(x0 + self.tile_size, width), min(y0 + self.tile_size, height)
                px0, px1 = max(x0 - self.tile_pad, 0), min(x1 + self.tile_pad, width)
                py0, py1 = max(y0 - self.tile_pad, 0), min(y1 + self.tile_pad, height)
                wins.append(((py0, py1, px0, px1), (y0, y1, x0, x1), (y0 - py0, x0 - px0)))
        return wins

    @staticmethod
    def shard_of(wins, rank, world):
        """Indices of the tiles rank `rank` computes: sort by input area (largest first), deal round-robin."""
        order = sorted(range(len(wins)), key=lambda i: (-(wins[i][0][1] - wins[i][0][0]) * (wins[i][0][3] - wins[i][0][2]), i))
        return [order[i] for i in range(rank, len(order), world)]

    def _run_tile(self, win):
        (py0, py1, px0, px1), (y0, y1, x0, x1), (oy, ox) = win
        s = self.scale
        with torch.no_grad():
            out = self.model(self.img[:, :, py0:py1, px0:px1])
        return out[:, :, oy * s:(oy + y1 - y0) * s, ox * s:(ox + x1 - x0) * s]

    def _run_tiles(self, wins, idxs):
        """{tile index: output core} for the tiles `idxs`; tiles with equal window size share batches of `tile_batch`."""
        groups = {}
        for i in idxs:
            (py0, py1, px0, px1) = wins[i][0]
            groups.setdefault((py1 - py0, px1 - px0), []).append(i)
        s, nb, done = self.scale, self.img.shape[0], {}
        for members in groups.values():
            for k in range(0, len(members), self.tile_batch):
                part = members[k:k + self.tile_batch]
                if len(part) == 1:
                    done[part[0]] = self._run_tile(wins[part[0]])
                    continue
                crops = torch.cat([self.img[:, :, wins[i][0][0]:wins[i][0][1], wins[i][0][2]:wins[i][0][3]] for i in part], 0)
                with torch.no_grad():
                    out = self.model(crops)
                for j, i in enumerate(part):
                    (y0, y1, x0, x1), (oy, ox) = wins[i][1], wins[i][2]
                    done[i] = out[j * nb:(j + 1) * nb, :, oy * s:(oy + y1 - y0) * s, ox * s:(ox + x1 - x0) * s]
        return done

    def tile_process(self):
        """Crop overlapping tiles, run each, paste the un-padded centres (utils.py:117-180)."""
        batch, channel, height, width = self.img.shape
        s = self.scale
        self.output = self.img.new_zeros((batch, channel, height * s, width * s))
        wins = self.tile_windows()
        import torch.distributed as dist
        world = dist.get_world_size() if (self.shard_tiles and dist.is_available() and dist.is_initialized()) else 1
        if world == 1:
            for i, t in self._run_tiles(wins, range(len(wins))).items():
                (y0, y1, x0, x1) = wins[i][1]
                self.output[:, :, y0 * s:y1 * s, x0 * s:x1 * s] = t
            return
        rank = dist.get_rank()
        mine = self.shard_of(wins, rank, world)
        done = {i: t.contiguous() for i, t in self._run_tiles(wins, mine).items()}
        if rank == 0:
            for i, t in done.items():
                (y0, y1, x0, x1) = wins[i][1]
                self.output[:, :, y0 * s:y1 * s, x0 * s:x1 * s] = t
            bufs, reqs = [], []
            for src in range(1, world):
                for i in self.shard_of(wins, src, world):
                    (y0, y1, x0, x1) = wins[i][1]
                    buf = self.img.new_empty((batch, channel, (y1 - y0) * s, (x1 - x0) * s))
                    reqs.append(dist.irecv(buf, src=src, tag=i))
                    bufs.append((i, buf))
            for r in reqs:
                r.wait()
            for i, buf in bufs:
                (y0, y1, x0, x1) = wins[i][1]
                self.output[:, :, y0 * s:y1 * s, x0 * s:x1 * s] = buf
        else:
            reqs = [dist.isend(done[i], dst=0, tag=i) for i in mine]
            for r in reqs:
                r.wait()

    def post_process(self):
        if self.mod_scale is not None:
            _, _, h, w = self.output.size()
            self.output = self.output[:, :, 0:h - self.mod_pad_h * self.scale, 0:w - self.mod_pad_w * self.scale]
        if self.pre_pad != 0:
            _, _, h, w = self.output.size()
            self.output = self.output[:, :, 0:h - self.pre_pad * self.scale, 0:w - self.pre_pad * self.scale]
        return self.output

    def _upscale(self, img):
        self.pre_process(img)
        if self.tile_size > 0:
            self.tile_process()
        else:
            self.process()
        out = self.post_process()
        out = out.data.squeeze().float().cpu().clamp_(0, 1).numpy()
        return np.transpose(out[[2, 1, 0], :, :], (1, 2, 0))

    @torch.no_grad()
    def enhance(self, img, outscale=None, alpha_upsampler='realesrgan'):
        """img: HxW[xC] numpy (BGR[A] / gray; uint8, uint16 or float) -> (upscaled numpy, img_mode) (utils.py:193-263)."""
        if outscale is not None and float(outscale) != float(self.scale):
            raise NotImplementedError('outscale != scale needs cv2.resize(LANCZOS4); resize the result yourself')
        img = img.astype(np.float32)
        if np.max(img) > 256:
            max_range = 65535
            print('\tInput is a 16-bit image')
        else:
            max_range = 255
        img = img / max_range
        alpha = None
        if len(img.shape) == 2:
            img_mode = 'L'
            img = np.stack([img, img, img], axis=-1)
        elif img.shape[2] == 4:
            img_mode = 'RGBA'
            alpha = img[:, :, 3]
            img = img[:, :, 2::-1]
            if alpha_upsampler == 'realesrgan':
                alpha = np.stack([alpha, alpha, alpha], axis=-1)
        else:
            img_mode = 'RGB'
            img = img[:, :, ::-1]
        output_img = self._upscale(img)
        if img_mode == 'L':
            output_img = _bgr2gray(output_img)
        if img_mode == 'RGBA':
            if alpha_upsampler == 'realesrgan':
                output_alpha = _bgr2gray(self._upscale(alpha))
            else:
                a = torch.from_numpy(np.ascontiguousarray(alpha))[None, None]
                output_alpha = F.interpolate(a, scale_factor=self.scale, mode='bilinear', align_corners=False)[0, 0].numpy()
            output_img = np.concatenate([output_img, output_alpha[..., None]], axis=-1)
        if max_range == 65535:
            output = (output_img * 65535.0).round().astype(np.uint16)
        else:
            output = (output_img * 255.0).round().astype(np.uint8)
        return output, img_mode


def _bgr2gray(img):
    return (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).astype(np.float32)
