"""Where a command's standard normals come from (reference: `torch.randn` at
/root/reference/src/pytorch_mppi/mppi.py:203, one
(K,T,nu) draw per command) and how they reach the kernels.

`Draws` is the part of `MPPI` that binds a draw to the problem block: injected noise (parity tests), rng="torch" --
torch.randn's own values, computed by the engine's launch straight into its sample-minor rows (`_torch_stream_fill`),
the next command's draw inside this command's K3 launch (draw-ahead) --, rng="torch-native", and the engine's Philox
generator in its three forms (on chip: no array at all; generator launch + rows in memory; inside K1).  Row buffers are
owned here (one per size, reused by every
command)."""
import ctypes as C
import os
import logging

import torch

from . import _native as N
from ._util import _DT, _ptr

# rng="torch": (device, K, T, nu) -> did csrc/noise_torch.hip reproduce torch.randn bit for bit
# (Draws._torch_stream_fill)
_TORCH_ROWS = {}


class Draws:
    """mixin of controller.MPPI (state: `rng`, `_injected`, `_zbuf*`, `_next_*`, `_pf_rows`, `_spill`, `last_draw`; see
    MPPI.__init__)"""

    # ------------------------------------------------------------------------------------------
    # noise plumbing
    # ------------------------------------------------------------------------------------------
    def inject_noise(self, z):
        """Queue standard-normal draws in the reference's layout (K,T,nu) (KMPPI: (K,S,nu)) for
        the next `command()` instead of drawing them -- "identical inputs" for parity checks."""
        self._injected.append(z)

    def _noise_shape(self):
        return (self.K_local, self.T, self.nu)

    def _zpitch(self):
        """Row pitch (samples) of this controller's TNK4 noise arrays (engine's choice, mppi_noise_pitch)."""
        key = (self.K_local, self.dtype)
        if getattr(self, "_zpitch_cache", (None, 0))[0] != key:
            self._zpitch_cache = (key, N.noise_pitch(self.K_local, _DT[self.dtype]))
        return self._zpitch_cache[1]

    def _zelems(self, Tn):
        """Elements of a TNK4 array for a (Tn, nu) sequence over this controller's samples."""
        return N.noise_rows4(Tn, self.nu) * self._zpitch() * 4

    def _row_buffer(self, n):
        """The TNK4 row array a command generates or converts its normals into.  ONE buffer per size,
        reused by every command (stream order makes that safe; the lazily materialised attributes only
        ever refer to the LAST command's rows): an allocation less per command."""
        buf = self._zbuf.get(n)
        if buf is None or buf.dtype != self.dtype:
            if len(self._zbuf) > 4:
                self._zbuf.clear()
            buf = self._zbuf[n] = torch.empty(n, device=self.d, dtype=self.dtype)
        return buf

    def _randn(self, *shape):
        # per-command sample draws; sharded torch modes draw from the shard's own generator
        return torch.randn(*shape, device=self.d, dtype=self.dtype, generator=self._shard_gen)

    def _draw_noise(self, p, shape):
        """Bind this command's standard normals to the problem: injected / torch.randn (reference
        layout, converted to the engine's sample-minor rows-of-4) or in-kernel Philox."""
        lib = N.lib()
        K, Tn, nu = shape
        self.last_draw = None
        if self.M > 1 and (self._injected or self.rng != "philox"):
            # the fused multi-rollout kernel keys its process-noise stream with the command number too
            self._call += 1
            p.call = self._call
        if self._injected:
            z = self._injected.pop(0)
            z = torch.as_tensor(z).to(device=self.d, dtype=self.dtype)
            if tuple(z.shape) == (self.K, Tn, nu) and self.K != K:
                z = z[self.k_offset:self.k_offset + K]          # global draw, this shard's rows
            if tuple(z.shape) != (K, Tn, nu):
                raise ValueError(f"injected noise has shape {tuple(z.shape)}, expected {(K, Tn, nu)}")
            z = z.contiguous()
        elif self.rng == "torch":
            if self.torch_rows and self._torch_stream_fill(p, K, Tn, nu):
                # the same values, already in the engine's rows
                return
            z = self._randn(K, Tn, nu)                                    # mppi.py:203
        elif self.rng == "torch-native":
            # same generator, drawn straight into the engine's sample-minor layout: no conversion
            # pass; which (k,t,n) gets which draw differs from the reference-layout draw
            zn = self._randn(self._zelems(Tn))
            p.noise_src = N.NOISE_TNK4
            p.z = _ptr(zn)
            p._keep["z"] = zn
            return
        else:
            self._call += 1
            p.noise_src = N.NOISE_PHILOX
            p.call = self._call
            p.z = None
            if self.philox_store and self._onchip_wanted(K, Tn, nu):
                self.last_draw = "philox-onchip"
                if self.onchip_spill:
                    # the rows that fit neither registers nor LDS wait for their sample's weight in this array (stored
                    # once, fetched once) instead of being generated a second time: 75.8 -> 71.3 us at C3
                    # (include/mppi_amd.h, ABI 20)
                    key = (K, Tn, nu)
                    sp = self._spill if self._spill is not None and self._spill[0] == key else None
                    if sp is None:
                        n = int(lib.mppi_onchip_spill_elems(C.byref(p)))
                        sp = self._spill = (key, torch.empty(n, device=self.d, dtype=self.dtype) if n > 0 else None)
                    if sp[1] is not None:
                        p.onchip_spill, p.onchip_spill_elems = _ptr(sp[1]), sp[1].numel()
                        p._keep["spill"] = sp[1]
                return
            if self.philox_store:
                # generate once, keep the rows for K3 to re-read: Philox + Box-Muller costs more per
                # element than an HBM read
                rows4 = N.noise_rows4(Tn, nu)
                n = self._zelems(Tn)
                # inside K1 every lane generates its own rows one after the other (~0.35 us per
                # row-of-4, however small K is); the generator launch spreads them over the whole chip
                # and costs one launch (~4 us): it wins from ~16 rows per sample on (tools/k_sweep.py)
                fill = self.philox_fill if self.philox_fill is not None else rows4 >= 16
                if self.M > 1 and not self._needs_generic():
                    fill = True            # the multi-rollout K1 reads its rows from memory
                self.last_draw = "philox-fill" if fill else "philox-k1"
                pf, self._pf_rows = self._pf_rows, None
                self._next_armed = None
                ahead = (self.draw_ahead and self.draw_ahead_philox and self.dtype == torch.float32
                        and (self._diagonal_sigma or not self.coloured_fill)
                         and self.d.type == "cuda" and not self._in_capture and K * Tn * nu >= self.draw_ahead_min)
                if pf is not None and pf[0] == (K, Tn, nu, int(p.k_offset), int(p.seed),
                        int(p.call)) and (fill or pf[2]):
                    # the rows of THIS command exist already: generated inside the previous command's K3 launch (ABI 21,
                    # csrc/noise_torch.hip -- the VALU that HBM-bound launch leaves idle; for small commands, the CUs)
                    # or while the previous command's collective ran.  Rows are a pure function of (seed, command,
                    # sample, row).
                    zn = pf[1]
                    self._pf_hits += 1
                    if pf[2]:
                        # the two row buffers change roles
                        self._zbuf_alt[n], self._zbuf[n] = self._zbuf.get(n), zn
                        self.last_draw = "philox-rows-ahead"
                    p.z = _ptr(zn)
                    p._keep["z"] = zn
                    p.noise_src = N.NOISE_TNK4
                    if ahead:
                        self._arm_next_philox(p, n)
                    return
                zn = self._row_buffer(n)
                p.z = _ptr(zn)
                p._keep["z"] = zn
                if ahead:
                    self._arm_next_philox(p, n)
                if fill:
                    # a separate generator launch at full occupancy (32 us for C3's 50 M normals, write
                    # floor 26 us), then K1 as the pure HBM-read kernel.  Short horizons keep the
                    # generation inside K1: one launch fewer.
                    if not self._diagonal_sigma and self.coloured_fill:
                        # full Sigma: the generator applies chol(Sigma) z + mu itself (full occupancy,
                        # a few us) and K1 / K3 run their diagonal form on the coloured rows instead of
                        # doing nu*(nu+1)/2 FMAs per timestep behind LDS reads at one wave per SIMD
                        rc = lib.mppi_noise_fill_philox_coloured(C.byref(p), p.z, self._stream())
                        if rc == 0:
                            p.noise_src, p.noise_coloured = N.NOISE_TNK4, 1
                            return
                        if rc != N.E_UNSUPPORTED:
                            N.check(rc, "mppi_noise_fill_philox_coloured")
                    N.check(lib.mppi_noise_fill_philox(C.byref(p), p.z, self._stream()), "mppi_noise_fill_philox")
                    p.noise_src = N.NOISE_TNK4
            return
        p._keep["z_ktn"] = z
        if self._ktn_direct_ok(p, Tn, nu, z):
            # fused fp32 path, diagonal Sigma: K1 and K3 read the reference-layout draw in place
            p.noise_src = N.NOISE_KTN
            p.z = _ptr(z)
            return
        self._convert_noise(p)

    def _torch_stream_fill(self, p, K, Tn, nu):
        """rng="torch": the values `torch.randn(K, Tn, nu)` would produce from the generator's present state, written by
        the engine's own launch straight into the rows K1 / K3 stream (csrc/noise_torch.hip, `mppi_noise_fill_torch`),
        and the generator advanced exactly as that call advances it -- every draw of the process, before and after, is
        what it would have been.  The first draw of every shape is compared with torch.randn itself, bit for bit, and
        the generator's offset with ATen's rule; a disagreement (another torch, another rocrand) switches this off for
        the process and the command draws with torch.randn as before.  False: not applicable here."""
        if (self.dtype != torch.float32 or (Tn * nu) % 4 or self.d.type != "cuda" or self._in_capture
                or _TORCH_ROWS.get("off") or torch.cuda.is_current_stream_capturing()):
            # (a capture: torch.randn registers its generator with the graph and replays advance it; the offset this
            # launch takes as an argument would be frozen -- capture_command() says so itself, a user's own
            # torch.cuda.graph() is caught by the query)
            return False
        gen = self._shard_gen if self._shard_gen is not None else torch.cuda.default_generators[self._dev_index]
        numel = K * Tn * nu
        cap = _TORCH_ROWS.get(("cap", self._dev_index))
        if cap is None:
            props = torch.cuda.get_device_properties(self._dev_index)
            cap = _TORCH_ROWS[("cap",
                    self._dev_index)] = props.multi_processor_count * (props.max_threads_per_multi_processor // 256)
        grid = min(cap, (numel + 255) // 256)
        inc = ((numel - 1) // (1024 * grid) + 1) * 4
        lib = N.lib()
        zn = self._row_buffer(self._zelems(Tn))
        pitch = self._zpitch()
        key = (self._dev_index, K, Tn, nu)
        if key not in _TORCH_ROWS:
            # once per shape and process: is this what torch.randn does here?
            state = gen.get_state()
            seed, off = gen.initial_seed(), gen.get_offset()
            ref = torch.randn(K, Tn, nu, device=self.d, dtype=self.dtype, generator=gen)
            moved = gen.get_offset() - off
            gen.set_state(state)
            rc = lib.mppi_noise_fill_torch(_ptr(zn), K, Tn, nu, pitch, seed, off, grid, self._stream())
            if rc == N.E_UNSUPPORTED:
                # a shape the launch does not take (more than 65535 rows-of-4): torch.randn
                _TORCH_ROWS[key] = False
                return False
            ok = rc == 0 and moved == inc
            if ok:
                rows = zn.view(-1, pitch, 4)[:, :K, :].permute(1, 0, 2).reshape(K, Tn, nu)
                ok = torch.equal(rows, ref)
            _TORCH_ROWS[key] = ok
            if not ok:
                import logging
                _TORCH_ROWS["off"] = True
                logging.getLogger("pytorch_mppi_amd").warning(
                    "pytorch_mppi_amd: torch.randn(%d, %d, %d) is not the stream csrc/noise_torch.hip reproduces (rc "
                            "%d, generator "
                    "offset +%d against +%d expected): rng='torch' keeps drawing with torch.randn", K, Tn, nu, rc,
                            moved, inc)
                return False
        elif not _TORCH_ROWS[key]:
            return False
        off, seed = gen.get_offset(), gen.initial_seed()
        nd, self._next_draw, self._next_armed = self._next_draw, None, None
        nkey = (K, Tn, nu, pitch, grid)
        if nd is not None and nd[0] == nkey and nd[1] is gen and nd[2] == seed and nd[3] == off:
            # command n-1's K3 launch generated exactly this draw beside its row stream (ABI 21, csrc/noise_torch.hip):
            # the
            # generator is where that launch assumed it would be -- same seed, same offset: the same values, by
            # construction. The two row buffers change roles
            n_el = self._zelems(Tn)
            self._zbuf_alt[n_el], self._zbuf[n_el] = zn, nd[4]
            zn = nd[4]
            self._next_hits += 1
            self._next_misses = 0
            self.last_draw = "torch-rows-ahead"
        else:
            if nd is not None:
                # generated for nothing: somebody else drew from the generator (reset(), the user's own randn)
                self._next_misses += 1
            N.check(lib.mppi_noise_fill_torch(_ptr(zn), K, Tn, nu, pitch, seed, off, grid, self._stream()),
                    "mppi_noise_fill_torch")
            self.last_draw = "torch-rows"
        gen.set_offset(off + inc)
        p.noise_src = N.NOISE_TNK4
        p.z = _ptr(zn)
        p._keep["z"] = zn
        self._next_cmds += 1
        if self.draw_ahead and numel >= self.draw_ahead_min and (self._next_misses < 2 or self._next_cmds % 64 == 0):
            # (a caller that draws from the generator between every two commands -- the reference's benchmark protocol
            # calls reset() -- makes every draw-ahead useless and K3 pays for it: after two misses in a row it is tried
            # only every 64th command)
            # ... and this command's K3 launch generates the NEXT draw -- the values torch.randn will produce from
            # (seed, off + inc) if nobody else draws from this generator in between -- into the other row buffer, on the
            # VALU the HBM-bound row stream leaves idle.  Whether the engine did (only the streaming diagonal K3 carries
            # it) is read back behind the command (_settle_next); whether the assumption held is checked above, at the
            # next command
            n_el = self._zelems(Tn)
            alt = self._zbuf_alt.get(n_el)
            if alt is None or alt.dtype != self.dtype or alt.device != zn.device:
                if len(self._zbuf_alt) > 2:
                    self._zbuf_alt.clear()
                alt = self._zbuf_alt[n_el] = torch.empty(n_el, device=self.d, dtype=self.dtype)
            p.next_z, p.next_seed, p.next_philox_offset = _ptr(alt), seed, off + inc
            p.next_grid_blocks, p.next_kind = grid, N.NEXT_DRAW_TORCH
            p._keep["next_z"] = alt
            self._next_armed = (nkey, gen, seed, off + inc, alt)
        return True

    def _arm_next_philox(self, p, n_el):
        """rng="philox", rows in memory: let this command's K3 launch generate the rows of the NEXT command (call + 1)
        into the other row buffer (MppiProblem.next_*, kind MPPI_NEXT_DRAW_PHILOX); _settle_next reads back whether it
        did"""
        alt = self._zbuf_alt.get(n_el)
        if alt is None or alt.dtype != self.dtype or alt.device != self.d or alt.data_ptr() == p.z:
            if len(self._zbuf_alt) > 2:
                self._zbuf_alt.clear()
            alt = self._zbuf_alt[n_el] = torch.empty(n_el, device=self.d, dtype=self.dtype)
        p.next_z, p.next_seed, p.next_philox_offset = _ptr(alt), int(p.seed), int(p.call) + 1
        p.next_grid_blocks, p.next_kind = 0, N.NEXT_DRAW_PHILOX
        p._keep["next_z"] = alt
        self._next_armed = ("philox", (int(p.K), int(p.T), int(p.nu), int(p.k_offset), int(p.seed), int(p.call) + 1),
                alt)

    def _settle_next(self, took=None):
        """behind the launches of a command: did its K3 generate the next draw (mppi_last_next_draw, thread-local --
        `took`:
        what the thread that issued the launches read there)?"""
        armed, self._next_armed = self._next_armed, None
        if armed is not None and (int(N.lib().mppi_last_next_draw()) if took is None else int(took)) == 1:
            if armed[0] == "philox":
                self._pf_rows = (armed[1], armed[2], True)
            else:
                self._next_draw = armed

    def _onchip_wanted(self, K, Tn, nu):
        """rng="philox": does this command go without a row array (include/mppi_amd.h, ABI 18; scope as checked again by
        the engine, csrc/rollout_onchip.hpp `onchip_problem_ok`)?"""
        if self.philox_onchip is False or self._onchip_refused:
            return False
        ok = (type(self).__name__ in ("MPPI", "SMPPI") and self.dtype == torch.float32 and self.M == 1
              and self.specific_action_sampler is None and Tn == self.T and not self._needs_generic()
              and self._model.model_id != N.MODEL_MLP)      # the dense MLP has its own matrix-core K1
        if not ok:
            return False
        if not self._diagonal_sigma:
            # a full Sigma CAN run on chip (L z + mu per timestep in the lane; csrc/rollout_onchip.hpp behind
            # MPPI_ONCHIP_FULL_SIGMA, tested at full size), but the factor rows come out of LDS every timestep and the
            # kernel becomes LDS-issue-bound: 0.127 ms at C3 against 0.104 ms for rows coloured by the generator launch
            # and streamed (profiles/r03_variants_philox.txt) -- not in the product build
            return False
        if self.philox_onchip:
            return True
        # On chip every lane generates its own rows one after the other (~0.35 us per row-of-4 however small K is): the
        # launch costs the same ~80 us at C3's horizon for K = 1024 and K = 65536, while the streaming form spreads the
        # generation over the chip.  Measured at T = 64, nu = 12 (tools/k_sweep.py, profiles/r03_k_sweep.txt against
        # r02_k_sweep.txt): K = 16384 0.083 vs 0.056 ms, K = 65536 0.087 vs 0.106, K >= 262144 8.0e8 vs 5.9e8 rollouts/s
        # ->
        # from three quarters of a full chip (one wave per SIMD = 65536 samples) upwards.  Round 6: where the two-wave kernel runs
        # (csrc/rollout_onchip_pair.hpp onchip_pair_model_ok: the integrator (16, 12) with a spill array) the launch costs ~60 us
        # and the streaming form passes it at K ~ 27000 (tools/onchip_threshold_sweep.py, profiles/r06_onchip_threshold_sweep.txt:
        # K = 28672 0.063 vs 0.064 ms, 32768 0.061 vs 0.068, 40960 0.062 vs 0.076)
        # (horizons from 40 steps on: below, every row fits registers + LDS of the one-wave kernel, no array, no two-wave kernel)
        two_wave = (self.onchip_spill and self._model.model_id == N.MODEL_INTEGRATOR and self.nx == 16 and nu == 12 and Tn >= 40
                    and os.environ.get("MPPI_ONCHIP_PAIR", "1") != "0")
        return K >= (28672 if two_wave else 49152)

    def _ktn_direct_ok(self, p, Tn, nu, z):
        return (self.ktn_direct and self.M == 1 and self.dtype == torch.float32 and self._diagonal_sigma
                and (Tn * nu) % 4 == 0
                and nu in (4, 8, 12, 16) and z.data_ptr() % 16 == 0 and p.num_envs <= 1 and Tn == self.T
                and not self._needs_generic())

    def _convert_noise(self, p):
        """(K,T,nu) draw kept in p._keep['z_ktn'] -> the engine's sample-minor rows-of-4."""
        z = p._keep["z_ktn"]
        K, Tn, nu = z.shape
        zn = self._row_buffer(self._zelems(Tn))
        N.check(N.lib().mppi_noise_from_ktn(C.byref(p), _ptr(z), _ptr(zn), self._stream()), "mppi_noise_from_ktn")
        p.noise_src = N.NOISE_TNK4
        p.z = _ptr(zn)
        p._keep["z"] = zn

    def _prefetch_philox_rows(self, p):
        """Sharded commands: the Philox rows of the NEXT command are a pure function of
        (seed, call+1, sample, row) -- nothing of this command's result enters -- so their generator
        launch is queued before the caller's stream waits for the record all-gather: the
        latency-bound collective (tens of microseconds over xGMI) hides behind 30 us of generation.
        The next command picks the buffer up if (shape, seed, call) still match, else drops it."""
        if p.noise_coloured:
            self._pf_rows = None
            return
        q = N.MppiProblem.from_buffer_copy(p)
        q.call = self._call + 1
        q.noise_src = N.NOISE_PHILOX
        n = self._zelems(q.T)
        zn = torch.empty(n, device=self.d, dtype=self.dtype)
        N.check(N.lib().mppi_noise_fill_philox(C.byref(q), _ptr(zn), self._stream()), "mppi_noise_fill_philox")
        self._pf_rows = ((q.K, q.T, q.nu, int(q.k_offset), int(q.seed), int(q.call)), zn, False)
