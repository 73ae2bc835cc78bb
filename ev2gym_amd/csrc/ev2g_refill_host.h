// ev2g_refill_host.h -- host side of ev2g_pool_refill (include/ev2g.h): checks that the config describes the shape of the resident pool,
// keeps device copies of the config's arrays while the config does not change, launches ev2g_refill_kernel (one wavefront per scenario)
// and the loader's table kernels for the refilled slots.  Everything is enqueued on the handle's stream; nothing is copied back.
// Included by ev2g_host.hip after the handle type.
#pragma once

static void refill_append(std::vector<unsigned char> &key, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    key.insert(key.end(), b, b + n);
}

static int ev2g_pool_refill_impl(ev2g_handle *h, const ev2g_gen_config *cfg, uint64_t seed, int64_t first_index, int32_t first_slot, int32_t n) {
    if (!h || !h->loaded) return fail(h, EV2G_ERR_STATE, "ev2g_pool_refill: no scenarios loaded");
    if (!cfg || n < 0 || first_slot < 0 || first_index < 0 || (long long)first_slot + n > h->M)
        return fail(h, EV2G_ERR_ARG, "ev2g_pool_refill: slot range outside the pool");
    if (!(h->cfg.flags & EV2G_FLAG_REFILLABLE) || h->sess_cap <= 0)
        return fail(h, EV2G_ERR_STATE, "ev2g_pool_refill: the pool was not loaded with EV2G_FLAG_REFILLABLE (fixed-size session blocks)");
    const DevScn &s = h->scn;
    const ev2g_gen_config &c = *cfg;
    // chargers with several ports, or a topology file: the kernel replays the reference's first-free port assignment per charger (RefillArgs::multi)
    const bool topo = c.topo_n_ports != nullptr;
    const bool multi = s.npc != 1 || s.het || topo;
    if (topo && !(c.topo_transformer && c.topo_phases && c.topo_min_charge_current && c.topo_max_charge_current && c.topo_min_discharge_current &&
                  c.topo_max_discharge_current && c.topo_voltage && c.topo_tr_max_power))
        return fail(h, EV2G_ERR_ARG, "ev2g_pool_refill: a topology needs all nine topo_* arrays");
    if (multi && (s.T > 256 || s.P > 256))
        return fail(h, EV2G_ERR_ARG, "ev2g_pool_refill: multi-port chargers / topology files are re-drawn on the device up to 256 steps and 256 ports (use ev2g_generate + ev2g_load_scenarios beyond)");
    long long cfg_ports = 0;
    for (int i = 0; i < c.number_of_charging_stations && i < (1 << 20); i++) cfg_ports += topo ? c.topo_n_ports[i] : c.number_of_ports_per_cs;
    if (c.simulation_length != s.T || c.timescale != s.dt || c.number_of_charging_stations != s.C || c.number_of_transformers != s.R ||
        cfg_ports != s.P || (!topo && c.number_of_ports_per_cs != s.npc))
        return fail(h, EV2G_ERR_ARG, "ev2g_pool_refill: the config does not describe the shape of the resident pool (steps, timescale, chargers, ports, transformers)");
    if (c.scenario < 0 || c.scenario > 2 || c.simulation_days < 0 || c.simulation_days > 2) return fail(h, EV2G_ERR_ARG, "ev2g_pool_refill: scenario / simulation_days out of range");
    if (c.n_ev_specs < 0 || (c.n_ev_specs > 0 && !(c.spec_registrations && c.spec_battery_capacity && c.spec_max_ac_charge_power && c.spec_max_ac_discharge_power)))
        return fail(h, EV2G_ERR_ARG, "ev2g_pool_refill: n_ev_specs > 0 needs the four spec_* model arrays");
    if ((c.tab_arrival_week || c.tab_arrival_weekend || c.tab_stay || c.tab_energy) && !(c.tab_arrival_week && c.tab_arrival_weekend && c.tab_stay && c.tab_energy))
        return fail(h, EV2G_ERR_ARG, "ev2g_pool_refill: the arrival / stay / energy tables of a data directory come together");
    const int nd = c.demand_response ? std::max(c.dr_events_per_day, 1) : 1;
    if (nd > s.ND) return fail(h, EV2G_ERR_ARG, "ev2g_pool_refill: more demand-response events per day than the resident pool has slots for");
    if (n == 0) return EV2G_OK;
    (void)hipSetDevice(h->device);

    // ---- device copies of the config's arrays: rebuilt only when the config (or the content of its arrays) changes ----
    std::vector<int> spec_row;
    const int n_spec_lut = ev2g_gen_spec_rows(c, spec_row);
    const int want_lut = c.n_ev_specs > 0 ? n_spec_lut : ((c.heterogeneous_ev_specs && c.fleet_with_efficiency_tables) ? EV2G_GEN_FLEET_MAX : 0);
    if (c.heterogeneous_ev_specs && want_lut != s.n_lut)
        return fail(h, EV2G_ERR_ARG, "ev2g_pool_refill: the config's fleet has " + std::to_string(want_lut) + " efficiency tables, the resident pool " +
                                         std::to_string(s.n_lut) + " (load the pool from ev2g_generate with the same config)");
    std::vector<unsigned char> key;
    refill_append(key, &c, sizeof c);
    const size_t ns = (size_t)std::max(c.n_ev_specs, 0);
    if (ns) {
        refill_append(key, c.spec_registrations, ns * 8); refill_append(key, c.spec_battery_capacity, ns * 8);
        refill_append(key, c.spec_max_ac_charge_power, ns * 8); refill_append(key, c.spec_max_ac_discharge_power, ns * 8);
        if (c.spec_efficiency) refill_append(key, c.spec_efficiency, ns * EV2G_LUT_LEN * 8);
    }
    if (c.tab_arrival_week) { refill_append(key, c.tab_arrival_week, 96 * 8); refill_append(key, c.tab_arrival_weekend, 96 * 8); refill_append(key, c.tab_stay, 48 * 8); refill_append(key, c.tab_energy, 48 * 8); }
    if (c.tab_pv && c.n_pv > 0) refill_append(key, c.tab_pv, (size_t)c.n_pv * 8);
    if (topo) refill_append(key, c.topo_tr_max_power, (size_t)s.R * 8);   // (the chargers' own constants are the resident pool's)
    refill_append(key, &h->load_gen, sizeof h->load_gen);   // (the dictionary entries below belong to the loaded pool)
    if (!h->d_refill_overflow) {   // its own allocation, freed by ev2g_destroy: it must survive ev2g_load_scenarios (which frees scn_allocs) and config changes
        HIPCHK(h, hipMalloc((void **)&h->d_refill_overflow, sizeof(int)));
        HIPCHK(h, hipMemsetAsync(h->d_refill_overflow, 0, sizeof(int), h->stream));
    }
    auto &rc_ = h->refill_cache;
    if (rc_.key != key) {
        (void)hipStreamSynchronize(h->stream);
        free_pool(rc_.allocs);
        rc_.key.clear();
        RefillArgs a{};
        a.cfg = c;
        a.cfg.topo_n_ports = nullptr; a.cfg.topo_transformer = nullptr; a.cfg.topo_phases = nullptr;
        int rc = 0;
        auto upd = [&](const double *src, size_t cnt, const double **dst) { *dst = nullptr; if (!src || !cnt) return 0; double *p; rc = upload(h, rc_.allocs, src, cnt, &p); *dst = p; return rc; };
        if (upd(c.spec_registrations, ns, &a.cfg.spec_registrations) || upd(c.spec_battery_capacity, ns, &a.cfg.spec_battery_capacity) ||
            upd(c.spec_max_ac_charge_power, ns, &a.cfg.spec_max_ac_charge_power) || upd(c.spec_max_ac_discharge_power, ns, &a.cfg.spec_max_ac_discharge_power) ||
            upd(c.spec_efficiency, c.spec_efficiency ? ns * EV2G_LUT_LEN : 0, &a.cfg.spec_efficiency) ||
            upd(c.tab_arrival_week, c.tab_arrival_week ? 96 : 0, &a.cfg.tab_arrival_week) || upd(c.tab_arrival_weekend, c.tab_arrival_week ? 96 : 0, &a.cfg.tab_arrival_weekend) ||
            upd(c.tab_stay, c.tab_arrival_week ? 48 : 0, &a.cfg.tab_stay) || upd(c.tab_energy, c.tab_arrival_week ? 48 : 0, &a.cfg.tab_energy))
            return rc;
        a.cfg.tab_pv = nullptr;   // the device uses the smoothed series below
        std::vector<double> pv;
        if (const char *err = ev2g_gen_pv_series(c, c.timescale, pv)) return fail(h, EV2G_ERR_ARG, err);
        if (upd(pv.data(), pv.size(), &a.pv_series)) return rc;
        a.tr_cap = nullptr;
        if (topo && upd(c.topo_tr_max_power, (size_t)s.R, &a.tr_cap)) return rc;
        if (!spec_row.empty()) { int *p; if ((rc = upload(h, rc_.allocs, spec_row.data(), spec_row.size(), &p))) return rc; a.spec_row = p; }
        ev2g_gen_make_run(c, s.P, s.npc, seed, a.g0);
        a.cls_of = nullptr;
        std::vector<int> cls_of;     // staging for asynchronous copies: these two live until the stream synchronisation below, like `pv`
        std::vector<ClsRec> tab;     // (indexed by entry; only the new ones are filled)
        if (s.dict) {
            // every (car model, charger) pair the generator can draw gets its dictionary entry now -- the operands exactly as the kernel's
            // write_session forms them (ev2g_refill.h) -- so that the device only looks an index up; entries the loaded batch did not
            // contain are appended to the resident dictionary
            Ev2gGenRun g = a.g0;
            g.c = &c;
            const Ev2gFleet fleet = ev2g_fleet(g);
            const int n_models = c.heterogeneous_ev_specs ? fleet.n : 1;
            const Ev2gRng rng0 = ev2g_rng(seed, 0);
            cls_of.resize((size_t)n_models * s.C);
            const size_t n_before = h->cls_map.size();
            for (int m = 0; m < n_models; m++) {
                Ev2gGenSession e{0, 1, 2, m, c.heterogeneous_ev_specs ? fleet.battery(m) : c.ev_battery_capacity,
                                 c.heterogeneous_ev_specs ? fleet.pac(m) : c.ev_max_ac_charge_power, 1.0};
                const Ev2gSessFields f = ev2g_gen_session_fields(g, rng0, e, spec_row.empty() ? nullptr : spec_row.data());
                for (int cs = 0; cs < s.C; cs++) {
                    const int ph = h->cs_ph_host[(size_t)cs];
                    const double v_gate = h->cs_vk_host[(size_t)cs * 4 + ph];
                    SessRec r{};
                    r.B = e.B; r.minB = f.minB; r.emerg = f.min_emerg; r.pacmax = e.pac; r.pdismax = f.pdis_max; r.tsm = f.tsm;
                    r.gate_ch = f.pac_min * 1000.0 / v_gate;
                    r.gate_dis = f.pdis_min * 1000.0 / v_gate;
                    r.v = h->cs_vk_host[(size_t)cs * 4 + std::min(ph, f.phases)];
                    r.rB = 1.0 / r.B; r.rv = 1.0 / r.v;
                    const int k = cls_find_or_add(h->cls_map, tab, ev2g_cls_of(r));
                    if (k < 0) {   // the entries this call added never reach the device: take them out of the host mirror again
                        for (auto it = h->cls_map.begin(); it != h->cls_map.end();) it = ((size_t)it->second >= n_before) ? h->cls_map.erase(it) : std::next(it);
                    }
                    if (k < 0) return fail(h, EV2G_ERR_ARG, "ev2g_pool_refill: the fleet would take the battery-maths dictionary beyond its " + std::to_string(EV2G_CLS_CAP) +
                                                                " entries (load the pool with EV2G_NO_DICT=1)");
                    cls_of[(size_t)m * s.C + cs] = k;
                }
            }
            if (h->cls_map.size() > n_before) {
                HIPCHK(h, hipMemcpyAsync(h->d_cls_rec + n_before, tab.data() + n_before, (h->cls_map.size() - n_before) * sizeof(ClsRec), hipMemcpyHostToDevice, h->stream));
                h->scn.n_cls = (int)h->cls_map.size();
            }
            int *p; if ((rc = upload(h, rc_.allocs, cls_of.data(), cls_of.size(), &p))) return rc; a.cls_of = p;
        }
        a.g0.c = nullptr;
        a.g0.pv_per_day = pv.empty() ? 0 : 1440 / c.timescale;
        (void)hipStreamSynchronize(h->stream);   // the staging vectors are temporaries
        rc_.args = a;
        rc_.key = key;
    }
    RefillArgs a = rc_.args;
    a.lut_rowmax = h->d_lut_rowmax;
    a.seed = seed; a.first_index = first_index; a.first_slot = first_slot; a.n = n; a.cap = h->sess_cap;
    a.overflow = h->d_refill_overflow;
    a.g0.seed = seed;
    a.head_tab = h->d_head_tab; a.head_nh = h->head_nh; a.step_tab = h->d_step_tab;
    if (c.demand_response && c.dr_events_per_day > 16 && (s.win_tab || h->d_head_tab))
        return fail(h, EV2G_ERR_ARG, "ev2g_pool_refill: at most 16 demand-response events per day");
    a.multi = multi ? 1 : 0;
    a.dbg = nullptr;
    if (std::getenv("EV2G_REFILL_STAMPS")) {   // development: cycle stamps of workgroup 0, printed at the next call
        static unsigned long long *d_dbg = nullptr;
        if (!d_dbg) { (void)hipMalloc((void **)&d_dbg, 32 * 8); (void)hipMemset(d_dbg, 0, 32 * 8); }
        else {
            unsigned long long v[32];
            (void)hipStreamSynchronize(h->stream);
            (void)hipMemcpy(v, d_dbg, 256, hipMemcpyDeviceToHost);
            if (v[16] | v[17] | v[18]) {   // (-DEV2G_RF_SUBSTAMPS: the setpoint phase in parts, summed over the calls so far)
                std::fprintf(stderr, "[ev2g] refill setpoint parts (cycles, summed): order %llu | batches %llu | weights %llu | sums %llu | loads %llu | accumulate %llu | median prep %llu | median %llu | step rows %llu | masks %llu\n",
                             v[16], v[17], v[18], v[19], v[20], v[21], v[22], v[23], v[24], v[25]);
                (void)hipMemset(d_dbg + 16, 0, 16 * 8);
            }
            std::fprintf(stderr, "[ev2g] refill: workgroup 0 runs %llu cycles; the middle workgroup starts %lld cycles after it and runs %llu; the last one starts %lld after and runs %llu\n",
                         v[6] - v[0], (long long)(v[10] - v[0]), v[11] - v[10], (long long)(v[8] - v[0]), v[9] - v[8]);
            std::fprintf(stderr, "[ev2g] refill stamps (cycles): prices %llu | step tables %llu | pass 1 %llu | pass 2 %llu | transformer series %llu | observation tables %llu | setpoints %llu\n",
                         v[1] - v[0], v[2] - v[1], v[3] - v[2], v[4] - v[3], v[7] - v[4], v[5] - v[7], v[6] - v[5]);   // (series / tables: of the LAST transformer)
        }
        a.dbg = d_dbg;
    }
    const size_t lds = ev2g_refill_lds_bytes(s.T, s.P, h->sess_cap, a.multi);
    if (lds > 160 * 1024) return fail(h, EV2G_ERR_ARG, "ev2g_pool_refill: the scenario's work arrays exceed the LDS");
    if (lds > 48 * 1024) HIPCHK(h, hipFuncSetAttribute((const void *)ev2g_refill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (std::getenv("EV2G_REFILL_STAMPS")) {   // development: how many of these one-wavefront workgroups a CU holds
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)ev2g_refill_kernel, 64, lds);
        std::fprintf(stderr, "[ev2g] refill: %zu bytes of LDS per workgroup, %d workgroups per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor), %d workgroups\n", lds, nb, n);
    }
    hipLaunchKernelGGL(ev2g_refill_kernel, dim3(n), dim3(64), lds, h->stream, s, h->st, a, h->d_ss_afap);
    HIPCHK(h, hipGetLastError());
    // (the observation tables of the refilled slots -- window, head and step table -- are rebuilt by the kernel itself, from LDS)
    HIPCHK(h, hipGetLastError());
    h->refilled = true;
    return EV2G_OK;
}

static long long ev2g_pool_refill_overflows_impl(ev2g_handle *h) {
    if (!h || !h->d_refill_overflow) return 0;
    int v = 0;
    (void)hipSetDevice(h->device);
    if (hipStreamSynchronize(h->stream) != hipSuccess) return -1;
    if (hipMemcpy(&v, h->d_refill_overflow, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}
