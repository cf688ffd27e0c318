"""Drop-in for particleFilter.py's `ParticleFilter` (the shark-state estimator robotSim.py:665-701 drives):

    pf = ParticleFilter(init_x_shark, init_y_shark, init_auv_list)
    particles = pf.create()                                   # :311-317
    particles = pf.create_and_update(particles)               # :277-282
    particles = pf.update_weights(particles, list_of_range_bearing)   # :285-310
    xy_mean = pf.particleMean(particles); pf.meanError(*xy_mean)      # :153-177

The particle list lives on the GPU (`ParticleList`; iterating it downloads `Particle` records with the
reference's attribute names x_p, y_p, v_p, theta_p, weight_p).  Random draws continue numpy's *global*
legacy RandomState exactly like the reference (whose `random` is numpy.random, particleFilter.py:8):
the state is handed to the kernel with np.random.get_state() and written back with set_state(), so
`np.random.seed(k)` before `create()` reproduces the reference's run.  Many filters at once:
`_pf_lib.FilterBatch`.
"""
import math

import numpy as np

from . import _lib, _pf_lib

def angle_wrap(ang):
    """particleFilter.py:18-33 (one rounded add per level, like the recursion)"""
    while not (-math.pi <= ang <= math.pi):
        if ang > math.pi:
            ang += (-2 * math.pi)
        elif ang < -math.pi:
            ang += (2 * math.pi)
        else:
            return None
    return ang


def velocity_wrap(velocity):
    while velocity > 5:
        velocity += -5
    return velocity


class Particle:
    """record view of one list position (reference attribute names)"""
    __slots__ = ("x_p", "y_p", "v_p", "theta_p", "weight_p")

    def __init__(self, x_p, y_p, v_p, theta_p, weight_p):
        self.x_p, self.y_p, self.v_p, self.theta_p, self.weight_p = x_p, y_p, v_p, theta_p, weight_p


class ParticleList:
    """The list `create` / `create_and_update` / `update_weights` hand around, resident on the GPU.
    Only the most recent list of a filter is live (the reference's older lists are garbage too)."""

    def __init__(self, owner):
        self._owner = owner
        self._rows = None
        self.mean = None

    def _download(self):
        if self._rows is None:
            self._owner._require_live(self)
            self._rows = self._owner._batch.particles()[0][0]
        return self._rows

    def __len__(self):
        return self._owner.number_of_particles

    def __iter__(self):
        for r in self._download().tolist():
            yield Particle(*r)

    def __getitem__(self, i):
        r = self._download()[i]
        return Particle(*r.tolist()) if r.ndim == 1 else [Particle(*q) for q in r.tolist()]

    def as_array(self):
        """[N,5] x_p, y_p, v_p, theta_p, weight_p"""
        return self._download().copy()


class ParticleFilter:
    def __init__(self, init_x_shark, init_y_shark, init_auv_list, number_of_particles=1000, device=0):
        self.x_shark = init_x_shark
        self.y_shark = init_y_shark
        self.auv_list = init_auv_list
        self.number_of_particles = number_of_particles
        self._ctx = _lib.Context(device)  # one handle per filter: a handle keeps one resident batch
        self._batch = None
        self._live = None

    # -- numpy global-stream hand-over
    @staticmethod
    def _np_state():
        st = np.random.get_state()
        return st, np.asarray(st[1], dtype=np.uint32), int(st[2])

    def _np_restore(self, st):
        mt, pos = self._batch.rng_state()
        np.random.set_state((st[0], mt[0], int(pos[0]), st[3], st[4]))

    def _require_live(self, particles):
        if particles is not self._live:
            raise ValueError("stale particle list: only the list returned by the last call is resident on the GPU")

    def _check(self):
        st, _ = self._batch.status()
        if st[0] == 1:
            raise RecursionError("angle_wrap recursed deeper than CPython allows (or met nan)")
        if st[0] == 2:
            raise ValueError("a must be greater than 0 (empty list_of_new_particles)")

    def _new_list(self):
        self._live = ParticleList(self)
        return self._live

    def create(self):
        st, mt, pos = self._np_state()
        self._batch = _pf_lib.FilterBatch(self._ctx, 1, self.number_of_particles)
        self._batch.create([[self.x_shark, self.y_shark]], mt, pos)
        self._np_restore(st)
        return self._new_list()

    def create_and_update(self, particles):
        self._require_live(particles)
        st, mt, pos = self._np_state()
        self._batch.set_rng(mt, pos)
        self._batch.run(phases=_pf_lib.UPDATE, n_steps=1)
        self._np_restore(st)
        self._check()
        return self._new_list()

    def update_weights(self, particles, list_of_range_bearing):
        self._require_live(particles)
        for row in list_of_range_bearing:
            print("measurement number in particleFilter ", row[5] if len(row) > 5 else None, " theta: ", row[2])
        meas = np.array([[float(v) for v in row[:5]] for row in list_of_range_bearing], dtype=np.float64)
        st, mt, pos = self._np_state()
        self._batch.set_rng(mt, pos)
        self._batch.run(meas=meas[None, None], shark_xy=[[[self.x_shark, self.y_shark]]],
                        phases=_pf_lib.WEIGHTS | _pf_lib.MEAN)
        self._np_restore(st)
        self._check()
        out = self._new_list()
        mean, _, _ = self._batch.estimates()
        out.mean = [float(mean[0, 0, 0]), float(mean[0, 0, 1])]
        return out

    def particleMean(self, new_particles):
        if isinstance(new_particles, ParticleList):
            self._require_live(new_particles)
            if new_particles.mean is None:
                self._batch.run(shark_xy=[[[self.x_shark, self.y_shark]]], phases=_pf_lib.MEAN, n_steps=1)
                mean, _, _ = self._batch.estimates()
                new_particles.mean = [float(mean[0, 0, 0]), float(mean[0, 0, 1])]
            return list(new_particles.mean)
        sum_x = sum_y = 0
        count = 0
        for particle in new_particles:
            sum_x += particle.x_p
            sum_y += particle.y_p
            count += 1
        return [sum_x / count, sum_y / count]

    def meanError(self, x_mean, y_mean):
        x_difference = x_mean - self.x_shark
        y_difference = y_mean - self.y_shark
        range_error = math.sqrt((x_difference ** 2) + (y_difference ** 2))
        print("error")
        print(range_error)
        return range_error

    def normalize(self, weights_list):
        """particleFilter.py:127-151 (host helper; update_weights runs it on the device)"""
        final_newlist = []
        for weights in weights_list:
            denominator = max(weights)
            final_newlist.append([(1 / denominator) * weight for weight in weights])
        summed = []
        for index in range(len(final_newlist[0])):
            new_weight = 0
            for weight in final_newlist:
                new_weight += weight[index]
            summed.append(new_weight)
        final_denominator = max(summed)
        return [(1 / final_denominator) * weight for weight in summed]

    def particle_coordinates(self, particles):
        return [[p.x_p, p.y_p, p.weight_p] for p in particles]
