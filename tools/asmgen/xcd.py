"""The one-launch plan for n = 65536 / 32768: persistent workgroups, every row's three roles on one XCD (credit / ticket
scheduler of fused_header) and the per-XCD scratch pool experiment (lifo_*)."""
import os

from . import state as cfg
from .arith import T

# ---- per-XCD scratch pool (FUSED_LIFO): 32 row slots per XCD, a free mask at ctl + 128 + 4 xcd.  A row's a' and b' live
# in two slots from its first forward role until every block product has LOADED them (signalled a few microseconds into
# the product, after its first barrier), c' in a third from the first product's store to the last inverse role's end.
# "Lowest free slot first" keeps the set of slots in use as small as the concurrency allows, so a slot is rewritten
# while its previous (dead, dirty) contents still sit in the L2 -- the lines are overwritten there instead of being
# written back.  Consumers read the scratch with `nt` loads: measured (tools/probes) to miss the L1 and see other CUs'
# stores, which a slot that is reused within a launch needs.
# Aux record of a job at record + 1024 + 16 slot: {1 + job, (a slot + 1) | (b slot + 1) << 8, c word, products that loaded};
# c word: 0 none, bit 31 = a product is allocating it, low byte = c slot + 1.
def lifo_mask_addr(em, dst_pair, ctl_pair, tmp):
    """dst = ctl + 128 + 4 * xcd"""
    R = em.raw
    lo = int(dst_pair[2:].split(":")[0])
    clo = int(ctl_pair[2:].split(":")[0])
    R("s_and_b32 %s, s98, 7" % tmp)
    R("s_lshl_b32 %s, %s, 2" % (tmp, tmp))
    R("s_add_u32 %s, %s, 128" % (tmp, tmp))
    R("s_add_u32 s%d, s%d, %s" % (lo, clo, tmp))
    R("s_addc_u32 s%d, s%d, 0" % (lo + 1, clo + 1))


def lifo_pop(em, name, vt, mask_pair, out, t0, t1, spin):
    """out = index of a free slot, now taken (one lane active); bounded"""
    R = em.raw
    L = em.lines.append
    R("s_mov_b32 %s, 0" % spin)
    L(".Lpop_%s:" % name)
    R("global_load_dword v%d, v%d, %s sc1" % (vt, cfg.V_ZERO, mask_pair))
    R("s_waitcnt vmcnt(0)")
    R("v_readfirstlane_b32 %s, v%d" % (t0, vt))
    R("s_cmp_lg_u32 %s, 0" % t0)
    R("s_cbranch_scc1 .Lpop_%s_try" % name)
    R("s_sleep 8")
    R("s_add_u32 %s, %s, 1" % (spin, spin))
    R("s_cmp_lt_u32 %s, 0x200000" % spin)
    R("s_cbranch_scc1 .Lpop_%s" % name)
    R("s_trap 2")                                        # the pool never refills: fail loudly
    L(".Lpop_%s_try:" % name)
    R("s_ff1_i32_b32 %s, %s" % (out, t0))
    R("s_lshl_b32 %s, 1, %s" % (t1, out))
    R("s_not_b32 %s, %s" % (t0, t1))
    R("v_mov_b32_e32 v%d, %s" % (vt, t0))
    R("global_atomic_and v%d, v%d, v%d, %s sc0" % (vt, cfg.V_ZERO, vt, mask_pair))
    R("s_waitcnt vmcnt(0)")
    R("v_readfirstlane_b32 %s, v%d" % (t0, vt))
    R("s_and_b32 %s, %s, %s" % (t0, t0, t1))
    R("s_cmp_lg_u32 %s, 0" % t0)
    R("s_cbranch_scc0 .Lpop_%s" % name)                  # somebody else took that slot first


def lifo_slot_addr(em, dst_lo, slot_sgpr, scr_pair, tmp, NB):
    """s[dst_lo:dst_lo+1] = scr + ((xcd * 32 + slot) << NB)"""
    R = em.raw
    slo = int(scr_pair[2:].split(":")[0])
    R("s_and_b32 %s, s98, 7" % tmp)
    R("s_lshl_b32 %s, %s, 5" % (tmp, tmp))
    R("s_add_u32 %s, %s, %s" % (tmp, tmp, slot_sgpr))
    R("s_lshr_b32 s%d, %s, %d" % (dst_lo + 1, tmp, 32 - NB))
    R("s_lshl_b32 s%d, %s, %d" % (dst_lo, tmp, NB))
    R("s_add_u32 s%d, s%d, s%d" % (dst_lo, dst_lo, slo))
    R("s_addc_u32 s%d, s%d, s%d" % (dst_lo + 1, dst_lo + 1, slo + 1))


def lifo_product_loaded(em, NV):
    """injected after the block product's first barrier: its a' / b' blocks are in registers.  The last product of the
    job to get here returns both slots to the pool.  s[96:97] = the job's aux record, s100 = the two slots' bits."""
    R = em.raw
    L = em.lines.append
    R("v_readfirstlane_b32 s42, v%d" % cfg.V_TID)
    R("s_cmp_lg_u32 s42, 0")
    R("s_cbranch_scc1 .Lvl_done")
    R("s_mov_b64 exec, 1")
    R("v_mov_b32_e32 v7, 1")
    R("global_atomic_add v7, v%d, v7, s[96:97] offset:12 sc0" % cfg.V_ZERO)
    R("s_load_dwordx2 s[46:47], s[0:1], 0x60")           # ctl
    R("s_waitcnt vmcnt(0) lgkmcnt(0)")
    R("v_readfirstlane_b32 s42, v7")
    R("s_cmp_eq_u32 s42, %d" % (NV - 1))
    R("s_cbranch_scc0 .Lvl_restore")
    lifo_mask_addr(em, "s[46:47]", "s[46:47]", "s42")
    R("v_mov_b32_e32 v7, s100")
    R("global_atomic_or v%d, v7, s[46:47]" % cfg.V_ZERO)
    L(".Lvl_restore:")
    R("s_mov_b64 exec, -1")
    L(".Lvl_done:")


def lifo_product_store(em, NB):
    """injected in front of the block product's stores: learn (or allocate) the job's c' slot, point S_CROW at this
    product's block of it.  Every wave runs it (no workgroup exchange needed): v[40:41] are free by now (b is consumed)."""
    R = em.raw
    L = em.lines.append
    R("s_load_dwordx2 s[46:47], s[0:1], 0x60")           # ctl
    R("s_load_dwordx2 s[52:53], s[0:1], 0x50")           # scratch pool
    R("s_mov_b64 exec, 1")
    R("v_bfrev_b32_e32 v40, 1")                          # 0x80000000
    R("global_atomic_or v40, v%d, v40, s[96:97] offset:8 sc0" % cfg.V_ZERO)
    R("s_waitcnt vmcnt(0) lgkmcnt(0)")
    R("v_readfirstlane_b32 s42, v40")
    R("s_cmp_eq_u32 s42, 0")
    R("s_cbranch_scc0 .Lcs_wait")
    # first product of the job to finish: take a slot, publish it
    lifo_mask_addr(em, "s[46:47]", "s[46:47]", "s43")
    lifo_pop(em, "c", 40, "s[46:47]", "s44", "s42", "s43", "s45")
    R("s_add_u32 s42, s44, 1")
    R("v_mov_b32_e32 v40, s42")
    R("global_atomic_or v%d, v40, s[96:97] offset:8" % cfg.V_ZERO)
    R("s_branch .Lcs_known")
    L(".Lcs_wait:")
    R("s_mov_b32 s45, 0")
    L(".Lcs_poll:")
    R("s_and_b32 s44, s42, 0xff")
    R("s_cmp_lg_u32 s44, 0")
    R("s_cbranch_scc1 .Lcs_have")
    R("s_sleep 2")
    R("global_load_dword v40, v%d, s[96:97] offset:8 sc1" % cfg.V_ZERO)
    R("s_waitcnt vmcnt(0)")
    R("v_readfirstlane_b32 s42, v40")
    R("s_add_u32 s45, s45, 1")
    R("s_cmp_lt_u32 s45, 0x200000")
    R("s_cbranch_scc1 .Lcs_poll")
    R("s_trap 2")
    L(".Lcs_have:")
    R("s_sub_u32 s44, s44, 1")
    L(".Lcs_known:")
    R("s_mov_b64 exec, -1")
    lifo_slot_addr(em, 20, "s44", "s[52:53]", "s42", NB)
    R("s_lshl_b32 s42, s89, 15")
    R("s_add_u32 s20, s20, s42")
    R("s_addc_u32 s21, s21, 0")                          # S_CROW: block s89 of the c' slot


def fused_header(em, PER_ROW, NV, NSW, CG_LOG):
    R = em.raw
    L = em.lines.append
    NB = cfg.PIPE_LOGN + 3                                    # log2 bytes of a row
    LI, LV, LF = NSW.bit_length() - 1, NV.bit_length() - 1, NSW.bit_length()   # log2 roles per job: inverse, product, forward
    Z = cfg.V_ZERO
    T = cfg.LDS_TICKET      # +0 kind, +4 ticket | +16 counter offset (0: none), +20 target, +24 credit offset, +28 amount | +32 id, +36 t0, +40 t1

    def lane0():
        R("s_mov_b64 exec, 1")

    def all_lanes():
        R("s_mov_b64 exec, -1")

    def poll(name, off_sgpr, want_sgpr):
        """wait until the dword at record + off_sgpr equals want_sgpr (normally true at once); bounded"""
        R("s_add_u32 s84, s72, %s" % off_sgpr)
        R("s_addc_u32 s85, s73, 0")
        R("s_mov_b32 s92, 0")
        L(".Lpoll_%s:" % name)
        R("global_load_dword v7, v%d, s[84:85] sc1" % Z)
        R("s_waitcnt vmcnt(0)")
        R("v_readfirstlane_b32 s91, v7")
        R("s_cmp_eq_u32 s91, %s" % want_sgpr)
        R("s_cbranch_scc1 .Lpoll_%s_done" % name)
        R("s_sleep 4")
        R("s_add_u32 s92, s92, 1")
        R("s_cmp_lt_u32 s92, s62")
        R("s_cbranch_scc1 .Lpoll_%s" % name)
        R("s_trap 2")                                    # an input that never completes: fail loudly, do not hang
        L(".Lpoll_%s_done:" % name)

    def take(kind, cdw, bias, nxt):
        """wave 0, lane 0 active: take one credit of counter cdw (effective value = stored + bias SGPR or 0), then a ticket"""
        R("v_mov_b32_e32 v12, 1")
        R("global_atomic_sub v12, v%d, v12, s[72:73] offset:%d sc0" % (Z, 4 * cdw))
        R("s_waitcnt vmcnt(0)")
        R("v_readfirstlane_b32 s84, v12")
        if bias:
            R("s_add_u32 s84, s84, %s" % bias)
        R("s_cmp_gt_i32 s84, 0")
        R("s_cbranch_scc1 .Ltook_%d" % kind)
        R("v_mov_b32_e32 v12, 1")
        R("global_atomic_add v%d, v12, s[72:73] offset:%d" % (Z, 4 * cdw))   # lost the race for the last credit: give it back
        R("s_branch %s" % nxt)
        L(".Ltook_%d:" % kind)
        R("v_mov_b32_e32 v12, 1")
        R("global_atomic_add v12, v%d, v12, s[72:73] offset:%d sc0" % (Z, 128 + 4 * cdw))
        R("s_waitcnt vmcnt(0)")
        R("v_readfirstlane_b32 s83, v12")
        R("s_mov_b32 s82, %d" % kind)
        R("s_branch .Ldecided")

    def stamp_t1(name):
        """trace: the role's inputs are ready (wave 0 keeps the stamp in LDS)"""
        R("v_readfirstlane_b32 s84, v%d" % cfg.V_TID)
        R("s_cmp_lg_u32 s84, 0")
        R("s_cbranch_scc1 .Lt1_%s" % name)
        R("s_memtime s[84:85]")
        lane0()
        R("s_waitcnt lgkmcnt(0)")
        R("v_mov_b32_e32 v8, s84")
        R("ds_write_b32 v%d, v8 offset:%d" % (Z, T + 40))
        all_lanes()
        L(".Lt1_%s:" % name)

    # scheduling domain of this workgroup: 2^s59 independent domains per XCD (own record, own jobs, own ring) keep the
    # atomic traffic per record line low; workgroups of an XCD join them round-robin.  s98 = domain = xcd + 8 * sub
    R("s_getreg_b32 s98, hwreg(HW_REG_XCC_ID, 0, 4)")
    R("s_and_b32 s98, s98, 7")
    R("s_load_dwordx16 s[56:71], s[0:1], 0x30")
    R("s_waitcnt lgkmcnt(0)")
    R("s_lshl_b32 s42, s98, 2")
    R("s_add_u32 s42, s42, 64")
    R("s_add_u32 s72, s68, s42")
    R("s_addc_u32 s73, s69, 0")                          # ctl + 64 + 4 xcd: workgroups of this XCD seen so far
    R("v_readfirstlane_b32 s74, v%d" % cfg.V_TID)
    R("s_cmp_lg_u32 s74, 0")
    R("s_cbranch_scc1 .Ldom_wait")
    lane0()
    R("v_mov_b32_e32 v7, 1")
    R("global_atomic_add v7, v%d, v7, s[72:73] sc0" % Z)
    R("v_mov_b32_e32 v8, 0")
    R("s_waitcnt vmcnt(0)")
    R("ds_write_b32 v%d, v7 offset:%d" % (Z, T))
    R("ds_write_b32 v%d, v8 offset:%d" % (Z, T + 16))    # no completion to publish yet
    R("s_waitcnt lgkmcnt(0)")
    all_lanes()
    L(".Ldom_wait:")
    R("s_barrier")
    R("ds_read_b32 v7, v%d offset:%d" % (Z, T))
    R("s_waitcnt lgkmcnt(0)")
    R("v_readfirstlane_b32 s74, v7")
    R("s_lshl_b32 s75, 1, s59")
    R("s_sub_u32 s75, s75, 1")
    R("s_and_b32 s74, s74, s75")                         # sub
    R("s_lshl_b32 s74, s74, 3")
    R("s_add_u32 s98, s98, s74")
    R("s_barrier")                                       # (wave 0 reuses the LDS word)
    R("s_branch .Lticket")
    L(".Lnext:")
    R("s_waitcnt vmcnt(0) lgkmcnt(0)")                   # this wave's stores are in the L2
    R("s_barrier")
    L(".Lticket:")
    R("s_load_dwordx16 s[56:71], s[0:1], 0x30")
    R("s_waitcnt lgkmcnt(0)")
    R("s_mul_i32 s42, s98, 0x11000")
    R("s_add_u32 s42, s42, 4096")                        # records 68 KiB apart (different memory channels)
    R("s_add_u32 s72, s68, s42")
    R("s_addc_u32 s73, s69, 0")                          # s[72:73]: this XCD's record
    R("s_add_u32 s93, s59, 3")                           # log2 of the number of domains
    R("s_lshl_b32 s42, 1, s93")
    R("s_sub_u32 s42, s42, 1")
    R("s_sub_u32 s99, s56, s98")
    R("s_add_u32 s99, s99, s42")
    R("s_lshr_b32 s99, s99, s93")                        # jobs of this domain: rows dom, dom + 8 D, ...  (rows >= 8 D checked by the host)
    R("v_readfirstlane_b32 s74, v%d" % cfg.V_TID)
    R("s_cmp_lg_u32 s74, 0")
    R("s_cbranch_scc1 .Lsched_done")                     # waves 1..3 wait at the barrier for wave 0's decision
    # ---- wave 0: publish the finished role (and the credits it releases), then find the next one
    R("s_memtime s[86:87]")
    lane0()
    R("ds_read_b128 v[8:11], v%d offset:%d" % (Z, T + 16))  # counter offset, target, credit offset, amount
    R("ds_read_b128 v[14:17], v%d offset:%d" % (Z, T + 32)) # id, t0, t1, -
    R("s_waitcnt lgkmcnt(0)")
    R("v_readfirstlane_b32 s74, v8")
    R("s_cmp_eq_u32 s74, 0")
    R("s_cbranch_scc1 .Lt_noflag")
    R("v_readfirstlane_b32 s75, v9")
    R("v_readfirstlane_b32 s76, v10")
    R("v_readfirstlane_b32 s77, v11")
    R("s_add_u32 s84, s72, s74")
    R("s_addc_u32 s85, s73, 0")
    R("v_mov_b32_e32 v12, 1")
    R("global_atomic_add v12, v%d, v12, s[84:85] sc0" % Z)  # the role just finished: one more "done"
    R("s_waitcnt vmcnt(0)")
    R("v_readfirstlane_b32 s78, v12")
    R("s_add_u32 s78, s78, 1")
    R("s_cmp_eq_u32 s78, s75")
    R("s_cbranch_scc0 .Lt_posted")                       # not the last role of its stage
    if cfg.FUSED_LIFO:
        R("v_readfirstlane_b32 s79, v17")                # (v[14:17] = id, t0, t1, slots to free)
        R("s_cmp_eq_u32 s79, 0")
        R("s_cbranch_scc1 .Lt_nofree")
        lifo_mask_addr(em, "s[84:85]", "s[68:69]", "s80")
        R("v_mov_b32_e32 v12, s79")
        R("global_atomic_or v%d, v12, s[84:85]" % Z)     # the inverse stage is complete: its c' slot returns to the pool
        L(".Lt_nofree:")
    R("s_cmp_eq_u32 s77, 0")
    R("s_cbranch_scc1 .Lt_posted")
    R("s_add_u32 s84, s72, s76")
    R("s_addc_u32 s85, s73, 0")
    R("v_mov_b32_e32 v12, s77")
    R("global_atomic_add v%d, v12, s[84:85]" % Z)        # the next stage of that job (or the slot's next job) may start
    L(".Lt_posted:")
    # optional trace record {ticket | kind << 28, t0, t1, t2} (low words of s_memtime), 16 B per role, 2^16 per XCD
    R("s_cmp_eq_u64 s[70:71], 0")
    R("s_cbranch_scc1 .Lt_noflag")
    R("v_mov_b32_e32 v12, 1")
    R("global_atomic_add v12, v%d, v12, s[72:73] offset:16 sc0" % Z)
    R("s_waitcnt vmcnt(0)")
    R("v_readfirstlane_b32 s74, v12")
    R("s_and_b32 s74, s74, 0xffff")
    R("s_lshl_b32 s75, s98, 16")
    R("s_or_b32 s74, s74, s75")
    R("s_lshl_b32 s74, s74, 4")
    R("s_add_u32 s74, s70, s74")
    R("s_addc_u32 s75, s71, 0")
    R("v_mov_b32_e32 v17, s86")
    R("global_store_dwordx4 v%d, v[14:17], s[74:75]" % Z)
    L(".Lt_noflag:")
    R("v_mov_b32_e32 v8, 0")
    R("v_mov_b32_e32 v9, s86")
    R("ds_write_b32 v%d, v8 offset:%d" % (Z, T + 16))    # nothing to publish until a role is set up
    R("ds_write_b32 v%d, v9 offset:%d" % (Z, T + 36))    # t0: this workgroup is free
    R("s_mov_b32 s88, 0")                                # polls so far
    R("s_lshl_b32 s79, 1, s60")                          # R
    R("s_min_u32 s79, s79, s99")
    R("s_lshl_b32 s79, s79, %d" % LF)                    # forward credits the host's zero stands for: min(R, jobs) * 2 NSW
    L(".Lsched:")
    R("global_load_dwordx4 v[8:11], v%d, s[72:73] sc1" % Z)   # sc1: device scope; plain and sc0 loads hit in the L1
    R("s_waitcnt vmcnt(0)")
    R("v_readfirstlane_b32 s77, v10")                    # inverse credits
    R("s_cmp_gt_i32 s77, 0")
    R("s_cbranch_scc0 .Lsee_v")
    take(3, 2, None, ".Lsee_v")
    L(".Lsee_v:")
    R("v_readfirstlane_b32 s76, v9")                     # product credits
    R("s_cmp_gt_i32 s76, 0")
    R("s_cbranch_scc0 .Lsee_f")
    take(0, 1, None, ".Lsee_f")
    L(".Lsee_f:")
    R("v_readfirstlane_b32 s75, v8")                     # forward credits (biased)
    R("s_add_u32 s75, s75, s79")
    R("s_cmp_gt_i32 s75, 0")
    R("s_cbranch_scc0 .Lsee_exit")
    take(1, 0, "s79", ".Lsee_exit")
    L(".Lsee_exit:")
    R("v_readfirstlane_b32 s78, v11")
    R("s_cmp_eq_u32 s78, 0")
    R("s_cbranch_scc1 .Lnothing")
    R("s_mov_b32 s82, 4")                                # every inverse role has been handed out: done
    R("s_mov_b32 s83, 0")
    R("s_branch .Ldecided")
    L(".Lnothing:")
    R("s_sleep 8")
    R("s_cmp_lt_u32 s88, 8")
    R("s_cbranch_scc1 .Lnothing_short")
    R("s_sleep 60")                                      # nothing for a while: poll every ~2 us
    L(".Lnothing_short:")
    R("s_add_u32 s88, s88, 1")
    R("s_cmp_lt_u32 s88, s62")
    R("s_cbranch_scc1 .Lsched")
    R("s_trap 2")                                        # nothing became ready for seconds: fail loudly, do not hang
    L(".Ldecided:")
    R("v_mov_b32_e32 v10, s82")
    R("v_mov_b32_e32 v11, s83")
    R("ds_write_b64 v%d, v[10:11] offset:%d" % (Z, T))
    R("s_waitcnt lgkmcnt(0)")
    all_lanes()
    L(".Lsched_done:")
    R("s_barrier")
    R("ds_read_b64 v[10:11], v%d offset:%d" % (Z, T))
    R("s_waitcnt lgkmcnt(0)")
    R("v_readfirstlane_b32 s42, v10")                    # kind: 0 product, 1 forward, 3 inverse, 4 exit
    R("v_readfirstlane_b32 s2, v11")                     # role number within its kind
    R("s_cmp_eq_u32 s42, 4")
    R("s_cbranch_scc0 .Lwork")
    R("S_EXIT")
    L(".Lwork:")
    # ---- job, slot and sub-index of the role; its completion record
    #      s74 job, s77 slot, s78 epoch, s89 sub-index; s75 byte offset of the counter to bump, s76 its value when the stage
    #      is complete, s80 the credit word that stage completion feeds, s81 how many credits
    R("s_cmp_eq_u32 s42, 3")
    R("s_cbranch_scc0 .Ldec_not_i")
    R("s_lshr_b32 s74, s2, %d" % LI)
    R("s_and_b32 s89, s2, %d" % (NSW - 1))
    R("s_mov_b32 s75, 8")
    R("s_mov_b32 s76, %d" % NSW)
    R("s_mov_b32 s80, 0")                                # -> forward credits of the job that reuses the slot ...
    R("s_lshl_b32 s81, 1, s60")
    R("s_add_u32 s81, s81, s74")
    R("s_cmp_lt_u32 s81, s99")                           # ... if there is one
    R("s_cselect_b32 s81, %d, 0" % (2 * NSW))
    R("s_add_u32 s43, s2, 1")
    R("s_lshl_b32 s83, s99, %d" % LI)
    R("s_cmp_eq_u32 s43, s83")                           # the XCD's last inverse role: tell the idle workgroups to leave
    R("s_cbranch_scc0 .Ldec_done")
    R("v_readfirstlane_b32 s43, v%d" % cfg.V_TID)
    R("s_cmp_lg_u32 s43, 0")
    R("s_cbranch_scc1 .Ldec_done")
    lane0()
    R("v_mov_b32_e32 v7, 1")
    R("global_atomic_add v%d, v7, s[72:73] offset:12" % Z)
    all_lanes()
    R("s_branch .Ldec_done")
    L(".Ldec_not_i:")
    R("s_cmp_eq_u32 s42, 0")
    R("s_cbranch_scc0 .Ldec_f")
    R("s_lshr_b32 s74, s2, %d" % LV)
    R("s_and_b32 s89, s2, %d" % (NV - 1))
    R("s_mov_b32 s75, 4")
    R("s_mov_b32 s76, %d" % NV)
    R("s_mov_b32 s80, 8")                                # -> inverse credits
    R("s_mov_b32 s81, %d" % NSW)
    R("s_branch .Ldec_done")
    L(".Ldec_f:")
    R("s_lshr_b32 s74, s2, %d" % LF)
    R("s_and_b32 s89, s2, %d" % (2 * NSW - 1))
    R("s_mov_b32 s75, 0")
    R("s_mov_b32 s76, %d" % (2 * NSW))
    R("s_mov_b32 s80, 4")                                # -> product credits
    R("s_mov_b32 s81, %d" % NV)
    L(".Ldec_done:")
    R("s_lshl_b32 s43, 1, s60")
    R("s_sub_u32 s43, s43, 1")
    R("s_and_b32 s77, s74, s43")                         # slot
    R("s_lshr_b32 s78, s74, s60")                        # epoch
    R("s_lshl_b32 s43, s77, 4")
    R("s_add_u32 s43, s43, 256")                         # the slot's counters
    R("s_add_u32 s75, s75, s43")
    R("s_add_u32 s83, s78, 1")
    R("s_mul_i32 s76, s76, s83")                         # the counter's value when this job's stage is complete
    R("v_readfirstlane_b32 s84, v%d" % cfg.V_TID)
    R("s_cmp_lg_u32 s84, 0")
    R("s_cbranch_scc1 .Lrec_done")
    lane0()
    R("v_mov_b32_e32 v8, s75")
    R("v_mov_b32_e32 v9, s76")
    R("v_mov_b32_e32 v10, s80")
    R("v_mov_b32_e32 v11, s81")
    R("ds_write_b128 v%d, v[8:11] offset:%d" % (Z, T + 16))
    R("s_lshl_b32 s84, s42, 28")
    R("s_and_b32 s85, s2, 0xfffffff")
    R("s_or_b32 s84, s84, s85")
    R("v_mov_b32_e32 v8, s84")
    R("ds_write_b32 v%d, v8 offset:%d" % (Z, T + 32))
    if cfg.FUSED_LIFO:
        R("v_mov_b32_e32 v8, 0")
        R("ds_write_b32 v%d, v8 offset:%d" % (Z, T + 44))  # pool slots to free when this role completes its stage (set by the inverse role)
    all_lanes()
    L(".Lrec_done:")
    # ---- the job's row: g = 8 D job + domain (modulus-major)
    R("s_add_u32 s84, s59, 3")
    R("s_lshl_b32 s84, s74, s84")
    R("s_add_u32 s84, s84, s98")
    R("s_mul_hi_u32 s3, s84, s58")                       # cm = g / batch
    R("s_mul_i32 s43, s3, s57")
    R("s_sub_u32 s86, s84, s43")                         # poly
    R("s_mul_i32 s87, s86, s14")
    R("s_add_u32 s87, s87, s3")                          # row = poly*nm + cm
    R("s_lshl_b32 s43, s3, %d" % (cfg.PIPE_LOGN + 4,))
    R("s_add_u32 s22, s10, s43")
    R("s_addc_u32 s23, s11, 0")                          # twiddles of the modulus
    R("s_lshr_b32 s83, s87, %d" % (32 - NB))
    R("s_lshl_b32 s82, s87, %d" % NB)                    # s[82:83]: byte offset of the row in the batch ...
    R("s_mov_b64 s[80:81], s[82:83]")                    # ... and in the scratch, which mirrors the batch (see above)
    R("s_lshl_b32 s43, s77, 4")
    R("s_add_u32 s79, s43, 256")                         # byte offset of the slot's counters in the record
    R("s_cmp_eq_u32 s42, 0")
    R("s_cbranch_scc1 .Lprep_v")
    R("s_cmp_eq_u32 s42, 3")
    R("s_cbranch_scc1 .Lprep_i")
    # ---- forward streaming role: operand s89 >> log NSW, column groups q = s89 mod NSW; the slot must be drained
    R("s_lshl_b32 s76, s78, %d" % LI)                    # inverse roles completed on the slot by earlier epochs
    R("s_add_u32 s75, s79, 8")
    poll("slot", "s75", "s76")
    if cfg.FUSED_LIFO:
        # the job's first forward role takes two slots from the XCD's pool and publishes them; everybody reads them
        R("s_lshl_b32 s43, s77, 4")
        R("s_add_u32 s43, s43, 1024")
        R("s_add_u32 s96, s72, s43")
        R("s_addc_u32 s97, s73, 0")                      # s[96:97]: the job's aux record
        R("s_add_u32 s76, s74, 1")                       # tag = job + 1
        R("s_cmp_lg_u32 s89, 0")
        R("s_cbranch_scc1 .Lf_slots")
        R("v_readfirstlane_b32 s43, v%d" % cfg.V_TID)
        R("s_cmp_lg_u32 s43, 0")
        R("s_cbranch_scc1 .Lf_slots")
        lane0()
        lifo_mask_addr(em, "s[84:85]", "s[68:69]", "s43")
        lifo_pop(em, "a", 7, "s[84:85]", "s80", "s42", "s43", "s81")
        lifo_pop(em, "b", 7, "s[84:85]", "s91", "s42", "s43", "s81")
        R("s_add_u32 s80, s80, 1")
        R("s_add_u32 s91, s91, 1")
        R("s_lshl_b32 s91, s91, 8")
        R("s_or_b32 s80, s80, s91")
        R("v_mov_b32_e32 v8, 0")
        R("v_mov_b32_e32 v9, 0")
        R("global_atomic_swap_x2 v%d, v[8:9], s[96:97] offset:8" % Z)   # c word, products that loaded
        R("v_mov_b32_e32 v7, s80")
        R("global_atomic_swap v%d, v7, s[96:97] offset:4" % Z)
        R("s_waitcnt vmcnt(0)")
        R("v_mov_b32_e32 v7, s76")
        R("global_atomic_swap v%d, v7, s[96:97]" % Z)    # the tag last: the record is valid for this job
        all_lanes()
        L(".Lf_slots:")
        R("s_mov_b32 s92, 0")
        L(".Lf_slots_poll:")
        R("global_load_dwordx2 v[10:11], v%d, s[96:97] sc1" % Z)
        R("s_waitcnt vmcnt(0)")
        R("v_readfirstlane_b32 s91, v10")
        R("v_readfirstlane_b32 s80, v11")
        R("s_cmp_eq_u32 s91, s76")
        R("s_cbranch_scc1 .Lf_slots_known")
        R("s_sleep 2")
        R("s_add_u32 s92, s92, 1")
        R("s_cmp_lt_u32 s92, s62")
        R("s_cbranch_scc1 .Lf_slots_poll")
        R("s_trap 2")
        L(".Lf_slots_known:")
        R("s_lshr_b32 s42, s89, %d" % LI)                # operand: 0 = a, 1 = b
        R("s_and_b32 s89, s89, %d" % (NSW - 1))
        R("s_lshl_b32 s43, s42, 3")
        R("s_lshr_b32 s80, s80, s43")
        R("s_and_b32 s80, s80, 0xff")
        R("s_sub_u32 s80, s80, 1")                       # the operand's slot
        lifo_slot_addr(em, 20, "s80", "s[64:65]", "s43", NB)
        R("s_lshl_b32 s43, s89, %d" % CG_LOG)            # the bytes of q column groups
        R("s_add_u32 s20, s20, s43")
        R("s_addc_u32 s21, s21, 0")
        R("s_add_u32 s82, s82, s43")                     # (no carry: the low bits were zero)
        R("s_cmp_eq_u32 s42, 0")
        R("s_cselect_b64 s[16:17], s[6:7], s[8:9]")
        R("s_add_u32 s16, s16, s82")
        R("s_addc_u32 s17, s17, s83")
    else:
        R("s_lshr_b32 s42, s89, %d" % LI)
        R("s_and_b32 s89, s89, %d" % (NSW - 1))
        R("s_lshl_b32 s43, s89, %d" % CG_LOG)                # the bytes of q column groups
        R("s_add_u32 s80, s80, s43")
        R("s_add_u32 s82, s82, s43")                         # (no carries: the low bits were zero)
        R("s_cmp_eq_u32 s42, 0")
        R("s_cselect_b64 s[16:17], s[6:7], s[8:9]")
        R("s_cselect_b64 s[20:21], s[64:65], s[66:67]")
        R("s_add_u32 s16, s16, s82")
        R("s_addc_u32 s17, s17, s83")
        R("s_add_u32 s20, s20, s80")
        R("s_addc_u32 s21, s21, s81")
    R("s_mov_b32 s90, 1")
    stamp_t1("f")
    R("s_branch .Lbody_f")
    L(".Lprep_i:")
    R("s_add_u32 s76, s78, 1")
    R("s_lshl_b32 s76, s76, %d" % LV)                    # every block product of the job
    R("s_add_u32 s75, s79, 4")
    poll("vdone", "s75", "s76")
    if cfg.FUSED_LIFO:
        R("s_lshl_b32 s43, s77, 4")
        R("s_add_u32 s43, s43, 1024")
        R("s_add_u32 s96, s72, s43")
        R("s_addc_u32 s97, s73, 0")
        R("global_load_dword v7, v%d, s[96:97] offset:8 sc1" % Z)       # the c word (published before any product completed)
        R("s_waitcnt vmcnt(0)")
        R("v_readfirstlane_b32 s80, v7")
        R("s_and_b32 s80, s80, 0xff")
        R("s_sub_u32 s80, s80, 1")
        R("v_readfirstlane_b32 s43, v%d" % cfg.V_TID)
        R("s_cmp_lg_u32 s43, 0")
        R("s_cbranch_scc1 .Li_free_noted")
        lane0()
        R("s_lshl_b32 s43, 1, s80")
        R("v_mov_b32_e32 v8, s43")
        R("ds_write_b32 v%d, v8 offset:%d" % (Z, T + 44))   # returned to the pool by whoever completes the inverse stage
        all_lanes()
        L(".Li_free_noted:")
        lifo_slot_addr(em, 16, "s80", "s[64:65]", "s43", NB)
        R("s_lshl_b32 s43, s89, %d" % CG_LOG)
        R("s_add_u32 s16, s16, s43")
        R("s_addc_u32 s17, s17, 0")
        R("s_add_u32 s82, s82, s43")
    else:
        R("s_lshl_b32 s43, s89, %d" % CG_LOG)
        R("s_add_u32 s80, s80, s43")
        R("s_add_u32 s82, s82, s43")
        R("s_add_u32 s16, s64, s80")
        R("s_addc_u32 s17, s65, s81")
    R("s_add_u32 s20, s4, s82")
    R("s_addc_u32 s21, s5, s83")
    R("s_mov_b32 s95, 2")
    stamp_t1("i")
    R("s_branch .Lbody_i")
    L(".Lprep_v:")
    R("s_add_u32 s76, s78, 1")
    R("s_lshl_b32 s76, s76, %d" % LF)                    # every forward role of the job
    R("s_mov_b32 s75, s79")
    poll("fdone", "s75", "s76")
    if cfg.FUSED_LIFO:
        R("s_lshl_b32 s43, s77, 4")
        R("s_add_u32 s43, s43, 1024")
        R("s_add_u32 s96, s72, s43")
        R("s_addc_u32 s97, s73, 0")                      # s[96:97]: the job's aux record (kept through the role)
        R("global_load_dword v7, v%d, s[96:97] offset:4 sc1" % Z)
        R("s_waitcnt vmcnt(0)")
        R("v_readfirstlane_b32 s80, v7")
        R("s_and_b32 s81, s80, 0xff")
        R("s_sub_u32 s81, s81, 1")                       # a' slot
        R("s_lshr_b32 s80, s80, 8")
        R("s_and_b32 s80, s80, 0xff")
        R("s_sub_u32 s80, s80, 1")                       # b' slot
        R("s_lshl_b32 s100, 1, s81")
        R("s_lshl_b32 s43, 1, s80")
        R("s_or_b32 s100, s100, s43")                    # both bits: returned to the pool once every product has loaded
        lifo_slot_addr(em, 16, "s81", "s[64:65]", "s43", NB)
        lifo_slot_addr(em, 18, "s80", "s[64:65]", "s43", NB)
        R("s_lshl_b32 s43, s89, 15")
        R("s_add_u32 s16, s16, s43")
        R("s_addc_u32 s17, s17, 0")
        R("s_add_u32 s18, s18, s43")
        R("s_addc_u32 s19, s19, 0")
        R("s_mov_b64 s[20:21], 0")                       # (the c' block is known when the stores start)
    else:
        R("s_lshl_b32 s43, s89, 15")
        R("s_add_u32 s80, s80, s43")
        R("s_add_u32 s16, s64, s80")
        R("s_addc_u32 s17, s65, s81")
        R("s_add_u32 s18, s66, s80")
        R("s_addc_u32 s19, s67, s81")
        R("s_mov_b64 s[20:21], s[16:17]")                    # the block product overwrites its a' block
    stamp_t1("v")
    R("s_branch .Lbody_v")
