"""`SimpleSparseUNet` (BACKBONES) — defined in the authors' mmdet3d fork [UNVENDORED]; configured at
projects/configs/nuScenes/FSF_nuScenes_config.py:58-70 and called at
projects/mmdet3d_plugin/models/detectors/single_stage_fsd.py:234.  Restated from the published SST module
(SURVEY.md App. C): SubM input conv, encoder stages (stride-2 SparseConv3d + SubMConv3d), decoder levels
(lateral SparseBasicBlock, concat, merge SubM, channel-reduce-add, SparseInverseConv3d / final SubM upsample),
output rows in the input voxel order.  Submodule names follow upstream (`conv_input`,
`encoder_layers.encoder_layerN`, `lateral_layerN`, `merge_layerN`, `upsample_layerN`)."""
import os

import collections

import torch
import torch.nn as nn

from .... import hip_ops, switches
from ...ops.spconv import SparseBasicBlock, SparseConvolution, SparseConvTensor, SparseSequential, make_sparse_convmodule
from ...registry import BACKBONES


@BACKBONES.register_module()
class SimpleSparseUNet(nn.Module):
    def __init__(self, in_channels, sparse_shape, order=("conv", "norm", "act"),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), base_channels=16, output_channels=128, ndim=3,
                 encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
                 decoder_channels=((64, 64, 64), (64, 64, 32), (32, 32, 16), (16, 16, 16)),
                 decoder_paddings=((1, 0), (1, 0), (0, 0), (0, 1)), keep_coors_dims=None, act_type="relu",
                 init_cfg=None):
        super().__init__()
        assert ndim == 3 and act_type == "relu"
        self.sparse_shape = list(sparse_shape)
        self.in_channels = in_channels
        self.order = tuple(order)
        self.base_channels = base_channels
        self.output_channels = output_channels
        self.encoder_channels = encoder_channels
        self.encoder_paddings = encoder_paddings
        self.decoder_channels = decoder_channels
        self.decoder_paddings = decoder_paddings
        self.stage_num = len(self.encoder_channels)
        self.keep_coors_dims = keep_coors_dims
        self.fp16_enabled = False
        assert isinstance(order, (list, tuple)) and len(order) == 3 and set(order) == {"conv", "norm", "act"}
        if self.order[0] != "conv":
            self.conv_input = make_sparse_convmodule(in_channels, base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                     indice_key="subm1", conv_type="SubMConv3d", order=("conv",))
        else:
            self.conv_input = make_sparse_convmodule(in_channels, base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                     indice_key="subm1", conv_type="SubMConv3d", order=self.order)
        enc_out = self.make_encoder_layers(norm_cfg, base_channels)
        self.make_decoder_layers(norm_cfg, enc_out)
        # plane-form outputs (K9c) are for layers whose consumer is another submanifold convolution: not the merge layers
        # (their output only meets the channel-reduced concat) nor the last upsample (its output goes to the neck)
        for lvl in range(1, self.stage_num + 1):
            getattr(self, f"merge_layer{lvl}")[0].emit_planes = False
        self.upsample_layer1[0].emit_planes = False
        # neighbour-mask row order of the fine levels (inference, see forward): the strided convolution INTO level l hands its
        # output rows over in that order and registers the level's submanifold rulebook under this key
        for lvl in range(2, self.stage_num + 1):
            first = list(getattr(self.encoder_layers, f"encoder_layer{lvl}")._modules.values())[0][0]
            first.reorder_output_key = f"subm{lvl}" if lvl <= switches.UNET_MASK_ORDER_LEVELS else None

    def make_encoder_layers(self, norm_cfg, in_channels):
        self.encoder_layers = SparseSequential()
        for i, blocks in enumerate(self.encoder_channels):
            blocks_list = []
            for j, out_channels in enumerate(tuple(blocks)):
                padding = tuple(self.encoder_paddings[i])[j]
                if i != 0 and j == 0:  # every stage but the first opens with a stride-2 sparse conv
                    blocks_list.append(make_sparse_convmodule(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2,
                                                              padding=padding, indice_key=f"spconv{i + 1}",
                                                              conv_type="SparseConv3d", order=self.order))
                else:
                    blocks_list.append(make_sparse_convmodule(in_channels, out_channels, 3, norm_cfg=norm_cfg,
                                                              padding=padding, indice_key=f"subm{i + 1}",
                                                              conv_type="SubMConv3d", order=self.order))
                in_channels = out_channels
            self.encoder_layers.add_module(f"encoder_layer{i + 1}", SparseSequential(*blocks_list))
        return out_channels

    def make_decoder_layers(self, norm_cfg, in_channels):
        block_num = len(self.decoder_channels)
        for i, block_channels in enumerate(self.decoder_channels):
            paddings = self.decoder_paddings[i]
            lvl = block_num - i
            setattr(self, f"lateral_layer{lvl}",
                    SparseBasicBlock(in_channels, block_channels[0],
                                     conv_cfg=dict(type="SubMConv3d", indice_key=f"subm{lvl}"), norm_cfg=norm_cfg))
            setattr(self, f"merge_layer{lvl}",
                    make_sparse_convmodule(in_channels * 2, block_channels[1], 3, norm_cfg=norm_cfg, padding=paddings[0],
                                           indice_key=f"subm{lvl}", conv_type="SubMConv3d", order=self.order))
            if lvl != 1:
                setattr(self, f"upsample_layer{lvl}",
                        make_sparse_convmodule(in_channels, block_channels[2], 3, norm_cfg=norm_cfg,
                                               indice_key=f"spconv{lvl}", conv_type="SparseInverseConv3d",
                                               order=self.order))
            else:  # the last level upsamples with a submanifold conv on the input sites
                setattr(self, f"upsample_layer{lvl}",
                        make_sparse_convmodule(in_channels, block_channels[2], 3, norm_cfg=norm_cfg, padding=paddings[1],
                                               indice_key="subm1", conv_type="SubMConv3d", order=self.order))
            in_channels = block_channels[2]

    @staticmethod
    def reduce_channel(x, out_channels):
        features = x.features
        n, in_channels = features.shape
        assert in_channels % out_channels == 0 and in_channels >= out_channels
        return x._like(features.view(n, out_channels, -1).sum(dim=2))

    def decoder_layer_forward(self, x_lateral, x_bottom, lateral_layer, merge_layer, upsample_layer, lateral_out=None):
        x = lateral_out if lateral_out is not None else lateral_layer(x_lateral)
        lat_planes, bot_planes = x.plane_sources, x_bottom.plane_sources
        f_bot, f_lat = x_bottom.features, x.features
        # cat((x_bottom.features, x.features), 1), written only if somebody reads it as one tensor
        x = x._like((f_bot, f_lat) if not torch.is_grad_enabled() else torch.cat((f_bot, f_lat), dim=1))
        if lat_planes is not None and len(lat_planes) == 1 and f_bot.size(1) <= 128 and f_bot.size(1) % 32 == 0:
            # the merge layer reads the concatenation as two plane sources: the lateral block's own plane-form output and
            # a conversion of the bottom-up features
            if bot_planes is None or len(bot_planes) != 1:
                bot_planes = [hip_ops.to_planes(f_bot)]
            x.plane_sources = [bot_planes[0], lat_planes[0]]
        x_merge = merge_layer(x)
        cout = x_merge.features.shape[1]
        no_grad = not (torch.is_grad_enabled() and (f_bot.requires_grad or f_lat.requires_grad or x_merge.features.requires_grad))
        if (no_grad and f_bot.size(1) + f_lat.size(1) == 2 * cout and cout % 4 == 0 and x._features is None
                and hip_ops.channel_pair_sum_add2_supported(f_bot, f_lat)):
            up = upsample_layer[0]
            add = x_merge.features
            if (switches.PLANES and f_bot.size(1) % 16 == 0 and f_lat.size(1) % 16 == 0 and f_bot.size(0) >= up.PLANES_MIN_ROWS
                    and getattr(up, "in_channels", 0) == cout and hip_ops.spconv_planes_supported([cout], up.out_channels, 27)
                    and up.kernel_size == [3, 3, 3]):
                # the sums' only reader is this level's upsampling convolution on K9d: they leave as planes (one launch instead of
                # sum + fsf_to_planes, the fp32 rows never written); anything that asks for `.features` gets them formed then
                y = x._like(None)
                y.plane_sources = [hip_ops.channel_pair_sum_add2_planes(f_bot, f_lat, add)]
                y.features_thunk = lambda: hip_ops.channel_pair_sum_add2(f_bot, f_lat, add=add)
                x = y
            else:
                x = x._like(hip_ops.channel_pair_sum_add2(f_bot, f_lat, add=add))  # reduce_channel + the add, no concatenation
        elif no_grad and x.features.is_cuda and x.features.dtype == torch.float32 and x.features.shape[1] == 2 * cout and cout % 4 == 0:
            x = x._like(hip_ops.channel_group_sum_add(x.features, cout, add=x_merge.features))  # reduce_channel + the add, one pass
        else:
            x = self.reduce_channel(x, cout)
            x = x._like(x_merge.features + x.features)
        return upsample_layer(x)

    @staticmethod
    def _plan_modules(meta, module):
        """The rulebooks `module` will ask for, built from coordinates alone (`meta` carries no features); returns the
        coordinate-only tensor of the module's output."""
        for m in module.modules():
            if isinstance(m, SparseConvolution) and not m.inverse:
                rb = m._rulebook(meta)
                if not m.subm:
                    meta = meta._like(None, rb.out_indices, rb.out_shape)
        return meta

    # Encoder levels `begin` runs.  Issued under the previous frame's tail (FSF.set_next_frame) the first levels' convolutions fill the
    # CUs its NMS / box-tail launches leave idle; same-box interleaved A/B on the 10-sweep frame (tools/profiling/frame_front_ab.py,
    # unannounced - announced): 0 levels (up to the input planes) 0.36-0.46 ms, 1 level 0.24-0.53, 2 levels 0.53-0.72, 3 levels 0.55-0.85,
    # 4 levels 0.59 — from the third level on the host waits of the index plan keep the frame's results waiting for nothing more
    BEGIN_LEVELS = 2

    def forward(self, voxel_info, batch_size=None):
        return self.finish(self.begin(voxel_info, batch_size))

    def begin(self, voxel_info, batch_size=None):
        """The first part of `forward`: the row order, the first levels' index plans (on the plan stream), the network's input in that
        order, `conv_input` and the first BEGIN_LEVELS encoder levels.
        `finish(begin(...))` is `forward`; a caller that knows the next frame early (FSF.set_next_frame) runs `begin` on a side
        stream under the previous frame's tail.  The streams `finish` runs on are the ones current THEN."""
        steps = self._forward_steps(voxel_info, batch_size)
        next(steps)
        return steps

    @staticmethod
    def finish(steps):
        try:
            next(steps)
        except StopIteration as done:
            return done.value
        raise RuntimeError("SimpleSparseUNet.finish: the forward did not end")

    def _forward_steps(self, voxel_info, batch_size=None):
        raw_coors = voxel_info["voxel_coors"]
        voxel_features = voxel_info["voxel_feats"]

        def own_coors():  # (launches: on whichever stream is current)
            c = raw_coors if self.keep_coors_dims is None else raw_coors[:, self.keep_coors_dims]
            return c.int().contiguous()

        if batch_size is None:
            batch_size = voxel_info.get("batch_size")
        if batch_size is None:
            batch_size = int(raw_coors[:, 0].max().item()) + 1  # upstream's host sync; callers that know B pass it
        # Row order INSIDE the network (inference): the plane kernels visit a (64-row block, kernel offset) pair only if a row of the
        # block has a neighbour there; with rows of equal neighbour mask adjacent (fsf_order_by_neighbor_mask: more neighbours first) a
        # block of the 0.2 m level touches 8.1 of 27 offsets instead of 16.8 and its live cells are 80 % full instead of 39 % (0.4 m
        # level: 24.8 -> 17.1, 61 -> 93 %).  The level's rows are permuted on the way in, every rulebook is built on the permuted
        # coordinates (so tables and features agree by construction), and the output goes back to the input order.
        reorder = (switches.UNET_MASK_ORDER and voxel_features.is_cuda and not torch.is_grad_enabled() and not self.training
                   and voxel_features.size(0) >= switches.UNET_MASK_ORDER_MIN_ROWS)
        # The index plan (inference): everything below that depends on the voxel COORDINATES only — the row order, every level's
        # rulebooks (hash build, strided proposal + sort with its host read-back, neighbour tables) — is a chain of ~25 small
        # launches per level.  On the main stream it stood between the levels' convolutions (1.2 ms per 10-sweep frame); here a
        # plan stream builds level k + 1's tables while the main stream runs level k's convolutions, and the first two levels' while
        # the voxel encoder is still at work (the coordinates carry the event of their creation).
        trace = getattr(self, "_trace", None)  # profiling hook (tools/profiling/plan_stream_trace.py): timed events on a stream

        def mark(tag, stream=None):
            if trace is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream if stream is not None else torch.cuda.current_stream())
                trace.append((tag, e))

        mark("forward enters")
        # Tensors that cross streams (the plan stream's tables read by the main and lateral streams, encoder outputs read by the lateral
        # stream, lateral outputs read by the main one) are HELD until an event the main stream records behind its last use of them
        # has completed, then dropped.  `Tensor.record_stream` did the same job at a price nobody saw on the host: when such a tensor
        # dies the caching allocator records one event per (block, stream that used it) — ~100 marker packets per frame, ~50 of them on
        # the main queue between the U-Net's last kernel and the neck, 2.7 us each: the main stream stood for 150-160 us there
        # (tools/profiling/neck_stall_probe.py: 158 us with both side streams, 117 / 43 with one, 0 with none).
        held = []
        pending = self.__dict__.setdefault("_held_cross_stream", collections.deque())
        while pending:
            if pending[0][0] is None:  # a forward that raised before it recorded its event: wait for everything once, then let go
                torch.cuda.synchronize()
            elif not pending[0][0].query():
                break
            pending.popleft()
        entry = [None, held]
        pending.append(entry)
        plan_on = (switches.UNET_PLAN_STREAM and voxel_features.is_cuda and not torch.is_grad_enabled() and not self.training
                   and not torch.cuda.is_current_stream_capturing())
        inv_perm = None
        levels = list(self.encoder_layers._modules.values())
        # Inference: the lateral blocks of the FINE levels (two big submanifold convolutions each, independent of everything below
        # them) go to a side stream once the encoder has left those levels: they fill the CUs that the small deep levels — a few
        # dozen workgroups per launch — leave idle, instead of running alone after them.
        side_levels = 0
        if (voxel_features.is_cuda and not torch.is_grad_enabled() and not self.training and switches.UNET_LATERAL_STREAM
                and voxel_features.size(0) >= switches.UNET_LATERAL_MIN_ROWS and not torch.cuda.is_current_stream_capturing()):
            side_levels = min(switches.UNET_LATERAL_LEVELS, self.stage_num - 1)
            if getattr(self, "_lateral_stream", None) is None:
                self._lateral_stream = torch.cuda.Stream()
        if plan_on:
            main = torch.cuda.current_stream()
            if getattr(self, "_plan_stream", None) is None:
                self._plan_stream = torch.cuda.Stream()
            ps = self._plan_stream
            ready = getattr(raw_coors, "_fsf_ready_event", None)
            if ready is not None:
                ps.wait_event(ready)
            else:
                ps.wait_stream(main)
            published = set()

            def publish(meta, extra=()):
                """the plan stream's tensors are consumed (and outlive their Python owners) on the main and lateral streams"""
                ts = list(extra)
                for rb in meta.indice_dict.values():
                    if not isinstance(rb, bool):
                        ts += [rb.nbr, rb.nbr_inv, rb.in_indices, rb.out_indices]
                for t in ts:
                    if t is not None and id(t) not in published:
                        published.add(id(t))
                        held.append(t)
                ev = torch.cuda.Event()
                ev.record(ps)
                torch.cuda.current_stream().wait_event(ev)  # (the stream current NOW: `begin` may have run on another one)
                mark("plan published", ps)

            with torch.cuda.stream(ps):
                coors = own_coors()
                perm64 = None
                if reorder:
                    perm, inv_perm = hip_ops.order_by_neighbor_mask(coors, batch_size, self.sparse_shape)
                    perm64, inv_perm = perm.long(), inv_perm.long()
                    coors = coors.index_select(0, perm64)
                meta = SparseConvTensor(None, coors, self.sparse_shape, batch_size)
                meta.indice_dict["__mask_order__"] = reorder
                meta = self._plan_modules(self._plan_modules(meta, self.conv_input), levels[0])
            publish(meta, [coors, perm64, inv_perm])
            x = self._permuted_input(voxel_features, perm64, coors, batch_size)
            x.indice_dict = meta.indice_dict
        else:
            coors = own_coors()
            if reorder:
                perm, inv_perm = hip_ops.order_by_neighbor_mask(coors, batch_size, self.sparse_shape)
                perm64, inv_perm = perm.long(), inv_perm.long()
                coors = coors.index_select(0, perm64)
            x = self._permuted_input(voxel_features, perm64 if reorder else None, coors, batch_size)
            x.indice_dict["__mask_order__"] = reorder  # (the dict is shared by every tensor derived from x)
        mark("conv_input starts")
        x = self.conv_input(x)
        encode_features = []
        lateral_out, lateral_done = {}, {}
        for level, encoder_layer in enumerate(levels, start=1):
            mark(f"encoder level {level} starts")
            x = encoder_layer(x)
            encode_features.append(x)
            if level == min(self.BEGIN_LEVELS, len(levels)):
                yield  # ---- `begin` ends here
            if plan_on and level < len(levels):  # the next level's tables, while the main stream runs this level's convolutions
                with torch.cuda.stream(ps):
                    meta = self._plan_modules(meta, levels[level])
                publish(meta)
            if side_levels > 0 and level == side_levels:
                main = torch.cuda.current_stream()
                side = self._lateral_stream
                side.wait_stream(main)  # the encoder outputs of levels 1..side_levels exist
                with torch.cuda.stream(side):
                    for lv in range(side_levels, 0, -1):  # the decoder needs the deepest of them first
                        src = encode_features[lv - 1]
                        # allocated on the main stream, read by the side stream: the allocator must not hand the blocks to a later
                        # main-stream allocation before the side kernels are through with them (an exception between here and the
                        # decoder loop would otherwise free them for reuse)
                        held.extend([src.features] + [u for pl in (src.plane_sources or []) for u in (pl.data, pl.scales)])
                        y = getattr(self, f"lateral_layer{lv}")(src)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        lateral_out[lv], lateral_done[lv] = y, ev
        x = encode_features[-1]
        mark("decoder starts")
        try:
            for i in range(self.stage_num, 0, -1):
                lat = lateral_out.get(i)
                if lat is not None:
                    main = torch.cuda.current_stream()
                    main.wait_event(lateral_done[i])
                    # (allocated on the side stream, consumed on this one: held, see `held`)
                    held.extend([lat.features] + [u for pl in (lat.plane_sources or []) for u in (pl.data, pl.scales)])
                x = self.decoder_layer_forward(encode_features[i - 1], x, getattr(self, f"lateral_layer{i}"),
                                               getattr(self, f"merge_layer{i}"), getattr(self, f"upsample_layer{i}"), lateral_out=lat)
        except BaseException:
            if lateral_done:  # whatever happens in the decoder, the main stream ends behind the side stream's work
                torch.cuda.current_stream().wait_stream(self._lateral_stream)
            raise
        # (the decoder loop has waited for every `lateral_done` event, the last things the side stream ran: a `wait_stream` here
        # would record one more event on a stream that went idle a millisecond ago — waking its hardware queue for that marker held
        # the main stream for ~150 us in front of the neck, on the 1-sweep frame as on the 10-sweep one)
        mark("decoder done")
        # `begin` may have run on another stream than this one (FSF's frame front): what it allocated there and this stream's kernels
        # read — the encoder outputs the decoder takes, the tables of a forward without a plan stream, the row maps — must not go back
        # to THAT stream's allocator while they are queued here; held like every other cross-stream tensor of the forward
        if not torch.is_grad_enabled():  # (training never splits its forward over two streams, and must not keep its graph alive here)
            held.append((encode_features, lateral_out, x, coors, inv_perm, voxel_features, voxel_info))
        if held and voxel_features.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())  # (the main stream has waited for every lateral event: behind this, nobody reads them)
            entry[0] = ev
        else:
            pending.pop()
        out = x.features
        # every decoder stage ends in a ReLU (asserted in __init__): no row can equal the neck's negative padding value, which spares
        # Voxel2PointScatterNeck its `pts_mask.all()` reduction and the host wait on its result
        out._fsf_nonnegative = True
        if inv_perm is not None:
            # back to the caller's voxel order — on first read of "voxel_feats"; a consumer that gathers rows anyway (the neck: one
            # row per POINT) composes its index with `inv_perm` instead and the [voxels, C] copy is never made
            return [PermutedRowsOutput(out, inv_perm)]
        return [{"voxel_feats": out}]

    def _permuted_input(self, voxel_features, perm64, coors, batch_size):
        """The network's input in the (permuted) row order: as planes made straight through the permutation when `conv_input` runs on
        the plane kernel (the permuted fp32 rows are then never written), else the gathered rows."""
        if perm64 is None:
            return SparseConvTensor(voxel_features, coors, self.sparse_shape, batch_size)
        conv = self.conv_input[0]
        if (switches.PLANES and voxel_features.is_cuda and voxel_features.dtype == torch.float32 and not torch.is_grad_enabled()
                and voxel_features.size(0) >= conv.PLANES_MIN_ROWS and conv.subm and conv.in_channels <= 128 and conv.in_channels % 32 == 0
                and voxel_features.size(1) == conv.in_channels and hip_ops.spconv_planes_supported([conv.in_channels], conv.out_channels, 27)):
            x = SparseConvTensor(None, coors, self.sparse_shape, batch_size)
            x.plane_sources = [hip_ops.to_planes(voxel_features, row_index=perm64)]
            x.features_thunk = lambda: voxel_features.index_select(0, perm64)
            return x
        return SparseConvTensor(voxel_features.index_select(0, perm64), coors, self.sparse_shape, batch_size)


class PermutedRowsOutput(dict):
    """`{"voxel_feats": rows.index_select(0, row_map)}` whose gather happens on first read of the key: `permuted` = (rows, row_map) for a
    consumer that indexes the rows itself (VoteSegmentor.extract_feat hands them to the neck)."""

    def __init__(self, rows, row_map):
        super().__init__()
        self.permuted = (rows, row_map)

    def __missing__(self, key):
        if key != "voxel_feats":
            raise KeyError(key)
        rows, row_map = self.permuted
        out = rows.index_select(0, row_map)
        out._fsf_nonnegative = getattr(rows, "_fsf_nonnegative", False)
        self[key] = out
        return out

    def __contains__(self, key):
        return key == "voxel_feats" or super().__contains__(key)

    def get(self, key, default=None):
        return self[key] if key in self else default
