"""Motion_plan_state -- the record type every planner takes and returns.

Mirrors the reference's constructor and attribute names (path_planning/motion_plan_state.py:3-19;
gym_rrt/envs/motion_plan_state_rrt.py:3-21 adds rl_state_id) so objects built by reference callers
can be passed straight in, and objects returned here can be consumed by them (robotSim / gym envs
read .x .y .theta .traj_time_stamp .parent .path .length .cost).  Any object with those attributes
is accepted as input; this class is what the planners hand back.
"""


class Motion_plan_state:
    __slots__ = ("x", "y", "z", "theta", "v", "w", "traj_time_stamp", "plan_time_stamp", "size", "parent",
                 "path", "length", "cost", "rl_state_id")

    def __init__(self, x, y, z=0, theta=0, v=0, w=0, traj_time_stamp=0, plan_time_stamp=0, size=0, length=0,
                 rl_state_id=None):
        self.x = x
        self.y = y
        self.z = z
        self.theta = theta
        self.v = v
        self.w = w
        self.traj_time_stamp = traj_time_stamp
        self.plan_time_stamp = plan_time_stamp
        self.size = size
        self.parent = None
        self.path = []
        self.length = length
        self.cost = []
        self.rl_state_id = rl_state_id

    def _kind(self):
        timeless = self.traj_time_stamp == 0 and self.plan_time_stamp == 0
        still = self.theta == 0 and self.v == 0 and self.w == 0
        if self.z == 0 and still and timeless:
            return "xy"
        if still and self.size == 0 and timeless:
            return "xyz"
        if self.size != 0 and timeless:
            return "obstacle"
        if self.z == 0 and self.v == 0 and self.w == 0:
            return "dubins"
        return "full"

    def __repr__(self):
        k = self._kind()
        if k == "xy":
            return "MPS: [x=%s, y=%s]" % (self.x, self.y)
        if k == "xyz":
            return "MPS: [x=%s, y=%s, z=%s]" % (self.x, self.y, self.z)
        if k == "obstacle":
            return "MPS: [x=%s, y=%s, z=%s, size=%s]" % (self.x, self.y, self.z, self.size)
        if k == "dubins":
            return "MPS: [x=%s, y=%s, theta=%s, trag_time=%s, plan_time=%s]" % (
                self.x, self.y, self.theta, self.traj_time_stamp, self.plan_time_stamp)
        return "MPS: [x=%s, y=%s, z=%s, theta=%s, v=%s, w=%s, trag_time=%s, plan_time=%s]" % (
            self.x, self.y, self.z, self.theta, self.v, self.w, self.traj_time_stamp, self.plan_time_stamp)

    __str__ = __repr__
