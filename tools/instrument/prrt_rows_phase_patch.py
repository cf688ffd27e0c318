# temporary phase clocks of prrt_rows_kernel (one wavefront prints its totals with printf at the end of the launch)
import os
p=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))+'/auv_sim_amd/csrc/planner_rows_kernel.h'
s=open(p).read()
def rep(old,new,cnt=1):
    global s
    assert s.count(old)==cnt,(s.count(old),old[:80])
    s=s.replace(old,new)
rep("""  for (;;) {
    // ---------------------------------------------------------------- rows without an episode take the next one""","""  unsigned long long ph[8] = {0,0,0,0,0,0,0,0};
  unsigned long long tf = __builtin_amdgcn_s_memtime();
  int n_trips = 0;
#define PHR(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[i] += tn - tf; tf = tn; __builtin_amdgcn_sched_barrier(0); } while (0)
  for (;;) {
    PHR(6);
    // ---------------------------------------------------------------- rows without an episode take the next one""")
rep("""    if (__any(act)) {
      // ---------------------------------------------------------------- bucket + node choice (:186, :214-223)""","""    PHR(0);
    if (__any(act)) {
      n_trips++;
      // ---------------------------------------------------------------- bucket + node choice (:186, :214-223)""")
rep("""      // ---------------------------------------------------------------- steer (:251-289)
      double cx = 0.0, cy = 0.0, cth = 0.0, ctt = 0.0;""","""      PHR(1);
      // ---------------------------------------------------------------- steer (:251-289)
      double cx = 0.0, cy = 0.0, cth = 0.0, ctt = 0.0;""")
rep("""      // ---------------------------------------------------------------- check_collision_free (:435-458)
      // closed rectangle""","""      PHR(2);
      // ---------------------------------------------------------------- check_collision_free (:435-458)
      // closed rectangle""")
rep("""      // ---------------------------------------------------------------- accept: add_node_to_grid (:108-159)
      int me = -1;""","""      PHR(3);
      // ---------------------------------------------------------------- accept: add_node_to_grid (:108-159)
      int me = -1;""")
rep("""      // ---------------------------------------------------------------- connect_to_goal_curve_alt(mps_list[-1]) (:374-423)
      // from the LAST list node""","""      PHR(4);
      // ---------------------------------------------------------------- connect_to_goal_curve_alt(mps_list[-1]) (:374-423)
      // from the LAST list node""")
rep("""      if (act) { prev_n_arc = n_arc; stepped = true; step++; }
    }""","""      if (act) { prev_n_arc = n_arc; stepped = true; step++; }
      PHR(5);
    }""")
# print at the end: find the kernel's end -- after the main loop there is a final section; use the last closing of the kernel
i=s.rindex("}\n\n#undef en")
s=s[:i]+"""  if (blockIdx.x == 0 && threadIdx.x == 0)
    printf("prrt_rows phases (clocks per trip of one wavefront, %d trips): take/idle %llu, selection %llu, steer %llu, collision %llu, insert %llu, arc %llu, finish+loop %llu\\n",
           n_trips, ph[0] / (unsigned long long)(n_trips > 0 ? n_trips : 1), ph[1] / (unsigned long long)(n_trips > 0 ? n_trips : 1), ph[2] / (unsigned long long)(n_trips > 0 ? n_trips : 1),
           ph[3] / (unsigned long long)(n_trips > 0 ? n_trips : 1), ph[4] / (unsigned long long)(n_trips > 0 ? n_trips : 1), ph[5] / (unsigned long long)(n_trips > 0 ? n_trips : 1),
           ph[6] / (unsigned long long)(n_trips > 0 ? n_trips : 1));
"""+s[i:]
open(p,'w').write(s)
