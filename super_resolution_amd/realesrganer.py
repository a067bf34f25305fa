"""RealESRGANer -- drop-in for Real-ESRGAN-master/realesrgan/utils.py:14-313 of the reference.

Same constructor, attributes (`scale, tile_size, tile_pad, pre_pad, mod_scale, half, device, model,
img, output, mod_pad_h, mod_pad_w`) and methods (`dni, pre_process, process, tile_process,
post_process, enhance`), plus `PrefetchReader` / `IOConsumer` (utils.py:266-313).  Differences, all
behind the same interface:
  * `half=True` selects bf16 MFMA compute inside the network (parameters stay float32); the image
    tensors handed to the model are float32 either way;
  * cv2 is not required: the colour conversions of `enhance` are numpy (COLOR_BGR2GRAY = 0.114 B +
    0.587 G + 0.299 R) and the `outscale != scale` resize (cv2.INTER_LANCZOS4, utils.py:256-261) is
    `lanczos4_resize` below -- both PARITY UNPINNED (cv2 is absent from the build image; OpenCV's
    8-bit path interpolates in 11-bit fixed point, this one in float32, so results may differ by 1 LSB);
  * when torch.distributed is initialised with world_size > 1 and `shard_tiles=True`, tile_process
    splits the tiles over the ranks (sorted by window shape, one contiguous block per rank) and rank 0 gathers the pasted
    centres with ONE point-to-point message per peer (a flat stack of that peer's cores in shard_of()
    order; no tags, no dependence on execution order) -- tiles are independent, so there is no collective.  Only
    rank 0 holds the assembled image: on the other ranks `self.output` is None after tile_process and
    `enhance` returns `(None, img_mode)`;
  * `tile_batch` runs that many tiles of equal window size through the network as one batch (every statistic of the
    network is per sample: the result is bit for bit the one-by-one result).  Default None = automatic for the
    networks of this package: as many tiles (at most 18) as fit half of the free device memory, sized with the
    executor's own workspace formula -- a forward costs ~50 ms beyond its pixels, so one tile at a time, the
    reference's loop, takes 6.7 s where batches take 3.5 s on a 4096^2 image; `tile_batch=1` restores that loop and
    its memory bound;
  * a failed tile raises instead of pasting a stale tile (utils.py:158-162 latent bug);
  * on a GPU `enhance` moves only the INTEGER image over PCIe: the caller's uint8 / uint16 image is uploaded as it is and
    converted to float planes by a kernel (sr_image_to_planes), and clamp / colour order / x255 / round / cast of the result
    run in one kernel (sr_image_from_planes) ahead of a pinned device -> host copy of the uint8 / uint16 image (0.8 GB
    instead of 3.2 GB for a 16384^2 RGB result); bit-identical to the numpy steps, which remain for CPU devices.
"""
import math
import queue
import threading

import numpy as np
import torch
from torch.nn import functional as F


def _load_checkpoint(path, loc='cpu'):
    if path.startswith('https://'):
        raise RuntimeError('no network access: download %s yourself and pass the local path' % path)
    return torch.load(path, map_location=torch.device(loc), weights_only=True)


class RealESRGANer():

    def __init__(self, scale, model_path, dni_weight=None, model=None, tile=0, tile_pad=10, pre_pad=10, half=False,
                 device=None, gpu_id=None, shard_tiles=False, tile_batch=None):
        """The reference signature (utils.py:29-40) plus two keyword extensions: `shard_tiles` deals the tiles over the
        ranks of an initialised process group; `tile_batch` = n runs n tiles of equal window size through the network
        in one batch (every statistic of the network is per sample, so the result is the one-by-one result, but the
        executor workspace grows by that factor); None picks n from the free device memory (see the module docstring),
        1 is the reference's loop."""
        if model is None:
            raise ValueError('RealESRGANer needs the network (model=...)')
        self.scale, self.tile_size, self.tile_pad, self.pre_pad = scale, tile, tile_pad, pre_pad
        self.mod_scale = None
        self.half = half
        self.shard_tiles = shard_tiles
        self._tile_batch_arg = None if tile_batch is None else max(1, int(tile_batch))
        self.tile_batch = self._tile_batch_arg or 1   # the batch in force; re-derived per image when automatic
        self.img = self.output = None
        self.mod_pad_h = self.mod_pad_w = 0
        if device is not None:
            self.device = device
        else:
            name = 'cuda:%d' % gpu_id if gpu_id else 'cuda'   # gpu_id 0 / None: the current device, as the reference
            self.device = torch.device(name if torch.cuda.is_available() else 'cpu')
        if isinstance(model_path, (list, tuple)):
            if dni_weight is None or len(model_path) != len(dni_weight):
                raise AssertionError('model_path and dni_weight must be lists of the same length')
            state = self.dni(model_path[0], model_path[1], dni_weight)
        elif model_path is not None:
            state = _load_checkpoint(model_path)
        else:
            state = None
        if state is not None:
            model.load_state_dict(state['params_ema' if 'params_ema' in state else 'params'], strict=True)
        self.model = model.eval().to(self.device)
        if half:
            if hasattr(self.model, 'compute_dtype'):
                self.model.compute_dtype = torch.bfloat16
            else:
                self.model = self.model.half()
        self._pinned = self._pinned_in = None

    def dni(self, net_a, net_b, dni_weight, key='params', loc='cpu'):
        """Deep network interpolation (utils.py:77-86): every entry of `key` becomes w0 * a + w1 * b."""
        a, b = _load_checkpoint(net_a, loc), _load_checkpoint(net_b, loc)
        wa, wb = dni_weight
        a[key] = {name: wa * va + wb * b[key][name] for name, va in a[key].items()}
        return a

    def _model_dtype(self):
        return torch.float16 if (self.half and not hasattr(self.model, 'compute_dtype')) else torch.float32

    # ------------------------------------------------------------------ geometry (utils.py:88-191)
    def pre_process(self, img):
        """HWC float numpy -> self.img, a 1CHW tensor on the device: reflect pre-pad on the right / bottom, then the
        reflect mod-pad that scale 2 / 1 networks (pixel_unshuffle by 2 / 4) need (utils.py:88-111)."""
        chw = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).float()
        self._pre_process_planes(chw[None].to(self.device).to(self._model_dtype()))

    def _pre_process_planes(self, t):
        """The padding half of pre_process on a (1, C, H, W) tensor that already sits on the device."""
        if self.pre_pad:
            t = F.pad(t, (0, self.pre_pad, 0, self.pre_pad), 'reflect')
        self.mod_scale = {2: 2, 1: 4}.get(self.scale, self.mod_scale)
        if self.mod_scale is not None:
            h, w = t.shape[-2:]
            self.mod_pad_h = -h % self.mod_scale
            self.mod_pad_w = -w % self.mod_scale
            t = F.pad(t, (0, self.mod_pad_w, 0, self.mod_pad_h), 'reflect')
        self.img = t

    def process(self):
        # inference only (the reference runs this under enhance()'s @torch.no_grad, utils.py:193): a caller that steps
        # through pre_process / process / post_process by hand must not end up on the differentiable training path
        with torch.no_grad():
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
        """Indices of the tiles rank `rank` computes: tiles sorted by (input area, largest first; window shape; index),
        cut into `world` contiguous blocks of equal count.  Tiles of one window shape share a forward (batches_of), and
        every forward costs ~50 ms of fixed time on top of its pixels, so a rank should hold as few shapes as possible:
        at the 4096^2 / tile 512 / pad 10 geometry 6 of 8 ranks run ONE forward of 8 tiles and the pixel imbalance is
        under 3 % (the window areas differ by 4 % at most).  Dealing the area-sorted tiles round-robin (rounds 1-2)
        balanced pixels to 0.2 % but gave every rank 3-4 shapes: slowest rank 747 -> 647 ms at 8 ranks, 1264 -> 1174 ms
        at 4 (tools/ab_shard_policy.py, profiles/r03_e_shard_policy_ab.json).  Geometry only: every rank derives every
        rank's shard."""
        order = sorted(range(len(wins)), key=lambda i: (-(wins[i][0][1] - wins[i][0][0]) * (wins[i][0][3] - wins[i][0][2]),
                                                        wins[i][0][1] - wins[i][0][0], i))
        n = len(order)
        return order[rank * n // world:(rank + 1) * n // world]

    @classmethod
    def shard_layout(cls, wins, rank, world, batch, channel, scale):
        """Where rank `rank` packs its output cores for the gather: [(tile index, element offset, core shape)] in
        shard_of() order, and the total element count.  Sender and receiver both derive it from the geometry."""
        out, off = [], 0
        for i in cls.shard_of(wins, rank, world):
            (y0, y1, x0, x1) = wins[i][1]
            shape = (batch, channel, (y1 - y0) * scale, (x1 - x0) * scale)
            out.append((i, off, shape))
            off += math.prod(shape)
        return out, off

    def _run_tile(self, win):
        (py0, py1, px0, px1), (y0, y1, x0, x1), (oy, ox) = win
        s = self.scale
        with torch.no_grad():
            out = self.model(self.img[:, :, py0:py1, px0:px1])
        return out[:, :, oy * s:(oy + y1 - y0) * s, ox * s:(ox + x1 - x0) * s]

    AUTO_TILE_BATCH_CAP = 18       # beyond this a forward's fixed cost is < 5 % of it
    AUTO_TILE_BATCH_MEMORY = 0.5   # fraction of the free device memory the automatic batch may claim

    def auto_tile_batch(self, wins):
        """Tiles per forward when `tile_batch` was left to the library: the largest n <= AUTO_TILE_BATCH_CAP (and <= the
        number of tiles of the most common window shape) whose executor workspace, crops and outputs fit
        AUTO_TILE_BATCH_MEMORY of the free device memory.  1 for networks that cannot be sized (not this package's
        RRDBNet), on the CPU, and for a single tile."""
        sized = getattr(self.model, 'workspace_bytes', None)
        if sized is None or not self.img.is_cuda or len(wins) < 2:
            return 1
        shapes = {}
        for w in wins:
            key = (w[0][1] - w[0][0], w[0][3] - w[0][2])
            shapes[key] = shapes.get(key, 0) + 1
        (h, w), count = max(shapes.items(), key=lambda kv: (kv[1], kv[0]))
        dev = self.img.device
        free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        budget = self.AUTO_TILE_BATCH_MEMORY * free
        nb, ch = self.img.shape[:2]
        out_ch = getattr(self.model, 'num_out_ch', ch)
        elt = self.img.element_size()
        held = self.model.workspace_held() if hasattr(self.model, 'workspace_held') else 0
        n = max(1, min(self.AUTO_TILE_BATCH_CAP, count))
        while n > 1:
            ws = sized(n * nb, h, w, self.img.dtype)
            need = max(0, ws - held) + n * nb * (ch * h * w + out_ch * h * w * self.scale * self.scale) * elt
            if ws and need <= budget:
                break
            n -= 1
        # same number of forwards with even batches: 36 tiles at n = 27 are 27 + 9, at 18 they are 18 + 18 (less memory)
        forwards = -(-count // n)
        return -(-count // forwards)

    @staticmethod
    def batches_of(wins, idxs, tile_batch):
        """The forwards that cover the tiles `idxs`, in execution order: tiles of equal window size share batches of
        `tile_batch` (the execution order need not be the order of `idxs` -- the gather must not depend on it)."""
        groups = {}
        for i in idxs:
            (py0, py1, px0, px1) = wins[i][0]
            groups.setdefault((py1 - py0, px1 - px0), []).append(i)
        return [members[k:k + tile_batch] for members in groups.values() for k in range(0, len(members), tile_batch)]

    def _run_tiles(self, wins, idxs, sink=None):
        """Output cores of the tiles `idxs`, batched as batches_of() says.  With `sink` every core is handed to
        sink(index, tensor) as soon as it exists (nothing is kept), else a dict is returned."""
        s, nb, done = self.scale, self.img.shape[0], {}
        put = sink if sink is not None else done.__setitem__
        for part in self.batches_of(wins, idxs, self.tile_batch):
            if len(part) == 1:
                put(part[0], self._run_tile(wins[part[0]]))
                continue
            crops = torch.cat([self.img[:, :, wins[i][0][0]:wins[i][0][1], wins[i][0][2]:wins[i][0][3]] for i in part], 0)
            with torch.no_grad():
                out = self.model(crops)
            for j, i in enumerate(part):
                (y0, y1, x0, x1), (oy, ox) = wins[i][1], wins[i][2]
                put(i, out[j * nb:(j + 1) * nb, :, oy * s:(oy + y1 - y0) * s, ox * s:(ox + x1 - x0) * s])
        return done

    def tile_process(self):
        """Crop overlapping tiles, run each, paste the un-padded centres (utils.py:117-180)."""
        batch, channel, height, width = self.img.shape
        s = self.scale
        wins = self.tile_windows()
        self.tile_batch = self._tile_batch_arg or self.auto_tile_batch(wins)
        import torch.distributed as dist
        world = dist.get_world_size() if (self.shard_tiles and dist.is_available() and dist.is_initialized()) else 1
        rank = dist.get_rank() if world > 1 else 0

        def paste(i, t):
            (y0, y1, x0, x1) = wins[i][1]
            self.output[:, :, y0 * s:y1 * s, x0 * s:x1 * s] = t

        if rank == 0:
            self.output = self.img.new_zeros((batch, channel, height * s, width * s))
        if world == 1:
            self._run_tiles(wins, range(len(wins)), sink=paste)
            return
        # Gather: every peer packs the cores of its tiles into ONE flat buffer laid out in shard_of() order (the
        # offsets depend on the tile geometry only, never on the order the tiles were run in) and sends it as one
        # message once its tiles are done; rank 0 receives one message per peer AFTER its own tiles.  Message matching
        # therefore needs neither tags (RCCL ignores them) nor an agreed execution order, and no receive kernel sits in
        # front of rank 0's compute.
        mine = self.shard_of(wins, rank, world)
        # gloo moves host memory only: GPU cores then travel through the host (rehearsals of the N > 1 path on one GPU;
        # RCCL, the backend of every real multi-GPU run, sends device buffers directly)
        via_host = self.img.is_cuda and getattr(dist, 'get_backend', lambda: 'nccl')() == 'gloo'
        if rank == 0:
            self._run_tiles(wins, mine, sink=paste)
            stacks = []
            for src in range(1, world):
                layout, total = self.shard_layout(wins, src, world, batch, channel, s)
                stacks.append((layout, self.img.new_empty(total, device='cpu' if via_host else self.img.device)))
            ops = [dist.P2POp(dist.irecv, buf, src) for src, (_, buf) in enumerate(stacks, start=1) if buf.numel()]
            for r in (dist.batch_isend_irecv(ops) if ops else []):
                r.wait()
            for layout, buf in stacks:
                buf = buf.to(self.img.device)
                for i, off, shape in layout:
                    paste(i, buf[off:off + math.prod(shape)].view(shape))
        else:
            self.output = None  # only rank 0 assembles the image
            layout, total = self.shard_layout(wins, rank, world, batch, channel, s)
            place = {i: (off, shape) for i, off, shape in layout}
            stack = self.img.new_empty(total)

            def pack(i, t):
                off, shape = place[i]
                stack[off:off + math.prod(shape)].view(shape).copy_(t)
            self._run_tiles(wins, mine, sink=pack)
            if total:
                for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, stack.cpu() if via_host else stack, 0)]):
                    r.wait()

    def post_process(self):
        """Crop the mod-pad and the pre-pad off the upscaled image (utils.py:182-191); None on ranks without the image."""
        if self.output is None:
            return None
        cut_h = cut_w = 0
        if self.mod_scale is not None:
            cut_h, cut_w = self.mod_pad_h * self.scale, self.mod_pad_w * self.scale
        cut_h += self.pre_pad * self.scale
        cut_w += self.pre_pad * self.scale
        h, w = self.output.shape[-2:]
        self.output = self.output[:, :, 0:h - cut_h, 0:w - cut_w]
        return self.output

    def _to_host(self, t):
        """Device tensor -> numpy (same dtype) through a pinned staging buffer (a pageable .cpu() of a 16384^2 x 3 image
        costs seconds of host time)."""
        t = t.contiguous()
        if not t.is_cuda:
            return t.numpy()
        nbytes = t.numel() * t.element_size()
        if self._pinned is None or self._pinned.numel() < nbytes:
            self._pinned = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        stage = self._pinned[:nbytes].view(t.dtype).view(t.shape)
        stage.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return stage.numpy().copy()

    def _run(self):
        if self.tile_size > 0:
            self.tile_process()
        else:
            self.process()
        return self.post_process()

    def _upscale(self, img):
        """Host path (CPU devices / stand-in networks): HxWx3 RGB float in -> HxWx3 BGR float in [0, 1] out."""
        self.pre_process(img)
        out = self._run()
        if out is None:
            return None
        out = np.clip(self._to_host(out.data.squeeze(0).float()), 0.0, 1.0)
        return np.ascontiguousarray(out[::-1].transpose(1, 2, 0))  # RGB planes -> H x W x BGR

    # ---- device path of enhance(): only the INTEGER image crosses PCIe, in both directions ----------------------------
    def _upload(self, img):
        """The caller's image as it is (uint8 / uint16 / float32 HWC or HW) -> device tensor."""
        a = np.ascontiguousarray(img if img.ndim == 3 else img[:, :, None])
        if a.dtype not in (np.uint8, np.uint16, np.float32):
            a = a.astype(np.float32)
        nbytes = a.nbytes
        if self._pinned_in is None or self._pinned_in.numel() < nbytes:
            self._pinned_in = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        stage = self._pinned_in[:nbytes]
        stage.numpy()[:] = a.reshape(-1).view(np.uint8)
        return stage.to(self.device, non_blocking=True), a.dtype, a.shape

    def _planes_from(self, dev_img, np_dtype, shape, chmap, full):
        """(1, 3, H, W) float planes = channels `chmap` of the uploaded image / full (utils.py:197-216) -> pre-padded self.img."""
        from . import _lib as L
        h, w, c = shape
        planes = torch.empty(1, 3, h, w, dtype=torch.float32, device=self.device)
        kind = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1, np.dtype(np.float32): 2}[np.dtype(np_dtype)]
        with torch.cuda.device(self.device):
            L.check(L.lib().sr_image_to_planes(dev_img.data_ptr(), kind, h, w, c, chmap[0], chmap[1], chmap[2], float(full),
                                               planes.data_ptr(), L.stream_ptr(self.device)), 'sr_image_to_planes')
        self._pre_process_planes(planes)

    def _quantize_into(self, out, dst, c0, gray, sixteen):
        """clamp / colour / x full / round / cast of the upscaled planes `out` (1, 3, H, W) into channels c0.. of the HWC
        integer device image `dst` (utils.py:225-228, 251-254)."""
        from . import _lib as L
        t = out[0]
        if t.stride(2) != 1 or t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float().contiguous()
        with torch.cuda.device(self.device):
            L.check(L.lib().sr_image_from_planes(t.data_ptr(), L.dtype_code(t.dtype), t.shape[1], t.shape[2], t.stride(0), t.stride(1),
                                                 int(gray), int(sixteen), dst.data_ptr(), dst.shape[2], c0,
                                                 L.stream_ptr(self.device)), 'sr_image_from_planes')

    def _enhance_device(self, img, sixteen, full, img_mode, alpha_upsampler):
        """enhance() between the float conversion and the Lanczos resize, on the device.  Returns the integer HWC image
        (numpy) or None on ranks that do not hold the assembled image."""
        dev_img, np_dtype, shape = self._upload(img)
        chmap = {'L': (0, 0, 0), 'RGB': (2, 1, 0), 'RGBA': (2, 1, 0)}[img_mode]
        self._planes_from(dev_img, np_dtype, shape, chmap, full)
        out = self._run()
        nch = {'L': 1, 'RGB': 3, 'RGBA': 4}[img_mode]
        dst = None
        if out is not None:
            dst = torch.empty(out.shape[2], out.shape[3], nch, dtype=torch.uint16 if sixteen else torch.uint8, device=self.device)
            self._quantize_into(out, dst, 0, img_mode == 'L', sixteen)
        host_alpha = None
        if img_mode == 'RGBA':
            if alpha_upsampler == 'realesrgan':
                self._planes_from(dev_img, np_dtype, shape, (3, 3, 3), full)
                out_a = self._run()
                if out_a is not None:
                    self._quantize_into(out_a, dst, 3, True, sixteen)
            else:  # cv2.resize(INTER_LINEAR) of the reference: half-pixel-centre bilinear, on the host (a small plane)
                a = torch.from_numpy(np.ascontiguousarray(img[:, :, 3].astype(np.float32) / full))[None, None]
                host_alpha = F.interpolate(a, scale_factor=self.scale, mode='bilinear', align_corners=False)[0, 0].numpy()
        if dst is None:
            return None
        res = self._to_host(dst)
        if img_mode == 'L':
            res = res[:, :, 0]
        if host_alpha is not None:
            res[:, :, 3] = (host_alpha * float(full)).round().astype(res.dtype)
        return res

    @torch.no_grad()
    def enhance(self, img, outscale=None, alpha_upsampler='realesrgan'):
        """img: HxW[xC] numpy (BGR[A] / gray; uint8, uint16 or float) -> (upscaled numpy, img_mode) (utils.py:193-263)."""
        h_in, w_in = img.shape[:2]
        sixteen = np.max(img) > 256
        if sixteen:
            print('\tInput is a 16-bit image')
        full = 65535 if sixteen else 255
        img_mode = 'L' if img.ndim == 2 else ('RGBA' if img.shape[2] == 4 else 'RGB')
        if torch.device(self.device).type == 'cuda' and self._model_dtype() == torch.float32:
            out = self._enhance_device(img, sixteen, full, img_mode, alpha_upsampler)
            if out is None:
                return None, img_mode
        else:
            out = self._enhance_host(img, sixteen, full, img_mode, alpha_upsampler)
            if out is None:
                return None, img_mode
        if outscale is not None and float(outscale) != float(self.scale):
            out = lanczos4_resize(out, int(h_in * outscale), int(w_in * outscale))
        return out, img_mode

    def _enhance_host(self, img, sixteen, full, img_mode, alpha_upsampler):
        """The same steps with numpy on the host (CPU devices; the yardstick of the device path's test)."""
        img = img.astype(np.float32) / full
        alpha = None
        if img_mode == 'L':
            rgb = np.repeat(img[:, :, None], 3, axis=2)
        elif img_mode == 'RGBA':
            rgb, alpha = img[:, :, 2::-1], img[:, :, 3]
        else:
            rgb = img[:, :, ::-1]
        out = self._upscale(rgb)
        out_alpha = None
        if img_mode == 'RGBA':
            if alpha_upsampler == 'realesrgan':
                a3 = self._upscale(np.repeat(alpha[:, :, None], 3, axis=2))
                out_alpha = None if a3 is None else _bgr2gray(a3)
            else:  # cv2.resize(INTER_LINEAR) of the reference: half-pixel-centre bilinear
                a = torch.from_numpy(np.ascontiguousarray(alpha))[None, None]
                out_alpha = F.interpolate(a, scale_factor=self.scale, mode='bilinear', align_corners=False)[0, 0].numpy()
        if out is None:
            return None
        if img_mode == 'L':
            out = _bgr2gray(out)
        elif img_mode == 'RGBA':
            out = np.concatenate([out, out_alpha[..., None]], axis=-1)
        return (out * float(full)).round().astype(np.uint16 if sixteen else np.uint8)


def _bgr2gray(img):
    return (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).astype(np.float32)


def _lanczos4_taps(n_out, n_in):
    """Source indices (n_out, 8) and weights (n_out, 8) of OpenCV's INTER_LANCZOS4 along one axis: source coordinate
    (dst + 0.5) * n_in / n_out - 0.5, taps floor - 3 .. floor + 4, Lanczos a = 4 weights normalised to sum 1,
    indices clamped to the image (BORDER_REPLICATE, what cv2.resize uses)."""
    pos = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
    base = np.floor(pos)
    frac = pos - base
    offs = np.arange(-3, 5, dtype=np.float64)
    x = frac[:, None] - offs[None, :]
    with np.errstate(divide='ignore', invalid='ignore'):
        w = np.where(np.abs(x) < 1e-12, 1.0, np.sin(np.pi * x) * np.sin(np.pi * x / 4) / (np.pi * np.pi * x * x / 4))
    w = np.where(np.abs(x) < 4, w, 0.0)
    w = w / w.sum(axis=1, keepdims=True)
    idx = np.clip(base[:, None].astype(np.int64) + offs[None, :].astype(np.int64), 0, n_in - 1)
    return idx, w.astype(np.float32)


def lanczos4_resize(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LANCZOS4) restated in float32 (utils.py:256-261):
    separable 8-tap Lanczos, replicate border, result rounded and saturated to the input's integer dtype.
    PARITY UNPINNED: cv2 is not installed here and no reference fixture holds a resized image."""
    src = img.astype(np.float32)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    iy, wy = _lanczos4_taps(out_h, src.shape[0])
    ix, wx = _lanczos4_taps(out_w, src.shape[1])
    rows = np.einsum('hkwc,hk->hwc', src[iy], wy)                 # (out_h, W, C)
    out = np.einsum('hwkc,wk->hwc', rows[:, ix], wx)              # (out_h, out_w, C)
    if squeeze:
        out = out[:, :, 0]
    if np.issubdtype(img.dtype, np.integer):
        info = np.iinfo(img.dtype)
        out = np.clip(np.rint(out), info.min, info.max).astype(img.dtype)
    return out


def _read_image(path):
    """cv2.imread(path, cv2.IMREAD_UNCHANGED) stand-in on PIL: numpy HxW (gray), HxWx3 BGR or HxWx4 BGRA;
    `.npy` files are loaded as they are (any dtype)."""
    if str(path).endswith('.npy'):
        return np.load(path)
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im)
    if a.ndim == 3 and a.shape[2] >= 3:
        a = np.concatenate([a[:, :, 2::-1], a[:, :, 3:]], axis=2)
    return np.ascontiguousarray(a)


def _write_image(path, img):
    if str(path).endswith('.npy'):
        np.save(path, img)
        return
    from PIL import Image
    a = img
    if a.ndim == 3 and a.shape[2] >= 3:
        a = np.concatenate([a[:, :, 2::-1], a[:, :, 3:]], axis=2)
    Image.fromarray(np.ascontiguousarray(a)).save(path)


class PrefetchReader(threading.Thread):
    """Reads the images of `img_list` ahead of the consumer on a thread (utils.py:266-293); iterate to receive them.

    `read_fn(path) -> numpy` replaces cv2.imread (default: PIL / .npy, BGR channel order like cv2)."""

    def __init__(self, img_list, num_prefetch_queue, read_fn=None):
        super().__init__(daemon=True)
        self.que = queue.Queue(num_prefetch_queue)
        self.img_list = img_list
        self.read_fn = read_fn or _read_image

    def run(self):
        for path in self.img_list:
            self.que.put(self.read_fn(path))
        self.que.put(None)

    def __next__(self):
        item = self.que.get()
        if item is None:
            raise StopIteration
        return item

    def __iter__(self):
        return self


class IOConsumer(threading.Thread):
    """Writes results handed over a queue as {'output': numpy, 'save_path': str} until it receives 'quit'
    (utils.py:296-313).  `write_fn(path, img)` replaces cv2.imwrite (default: PIL / .npy)."""

    def __init__(self, opt, que, qid, write_fn=None):
        super().__init__(daemon=True)
        self._queue = que
        self.qid = qid
        self.opt = opt
        self.write_fn = write_fn or _write_image

    def run(self):
        while True:
            msg = self._queue.get()
            if isinstance(msg, str) and msg == 'quit':
                break
            self.write_fn(msg['save_path'], msg['output'])
        print(f'IO worker {self.qid} is done.')
