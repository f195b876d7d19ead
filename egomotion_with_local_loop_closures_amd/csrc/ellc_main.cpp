// ellc_main — a main.cpp-shaped driver over the facade (reference: src/main.cpp:199-505): tracks a sequence of
// grey frames against the active keyframe, refines / propagates the semi-dense depth map, switches keyframe every
// KEYFRAME_PROPAGATE_INTERVAL frames and writes the reference's result files:
//   poses_orig.txt   frameId kfId wx wy wz vx vy vz(poseWrtWorld) rescaleFactor seeds%        (main.cpp:373)
//   matchframes.txt  frameId kfId pose6(poseWrtOrigin) rescaleFactor seeds% 0 0 0               (main.cpp:382)
//   matchframes_globalopt.txt (LC mode)  testId matchId pose6 rescale seeds matchValue rms_error view_angle   (GlobalOptimize.cpp:580)
//   <id>_{Depth,Depth_pyr0,DepthVarArr_pyr0}.txt (--save-mats DIR)  text checkpoints of the active keyframe at every
//                    keyframe switch (ImageFunc.cpp:73-87, Frame.cpp:697-871); --replicate DIR reads them back (:58-66)
// --init-poses FILE  FLAG_INITIALIZE_NONZERO_POSE (main.cpp:207-225): one line "frameNo wx wy wz vx vy vz" (world pose,
//                    the so3poses7.txt of the rotation-averaging step) per tracked frame; supplies the initial rotation
// --no-fused         every stage of a tracked frame as its own call, the pose through the host (GetImagePoseEstimate, then observe /
//                    regularise / export). Default (--fused): tracked frames through ellc_track_frame — the depth stages are enqueued
//                    behind the alignment and read its pose on the device (same files, same bits); since r04 (fill + regularise +
//                    export in one launch behind the observation) 0.207-0.209 against 0.213-0.217 ms per frame in bench.py's loop
// --bgr              the input holds decoded full-size BGR frames (4W x 4H x 3 bytes each): grey conversion, undistortion with
//                    the reference's hard-coded camera (ExternVariable.h:53-62, scaled to the input size) and the 1/4
//                    resize run on the device (Frame.cpp:45-75); --no-undistort = FLAG_DO_UNDISTORTION off
// --world N --rank r [--device d] with --comm-id FILE (RCCL: rank 0 writes its 128-byte unique id there, the others read it)
//                    or --comm-tcp PORT (TCP on 127.0.0.1 through rank 0): one process per GPU, every process tracks the whole
//                    sequence, the loop-closure batch (GlobalOptimize.cpp:480-610) is sharded over the ranks by
//                    ellc_shard_range and its poses are gathered once per batch (ellc_gather_results); every rank writes
//                    the same files into its own out_dir
// Input is otherwise a header-less file of W*H u8 grey frames (the decode itself always stays outside).
// In LC mode finished keyframes go through the loop-closure ring (facade class globalOptimize): matching and the batched
// alignment of a pushed keyframe run on a second thread and a second context beside tracking, joined at the next push
// (GlobalOptimize.cpp:241 / :161); tracking-loss recovery
// (findConnection) and the MATLAB rotation averaging are not part of this path.
#include "../../include/ellc_facade.hpp"
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <unistd.h>

using namespace ellc;

int main(int argc, char** argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s frames.raw W H num_frames out_dir [LC] [levels] [--save-mats DIR] [--replicate DIR] [--init-poses FILE] [--bgr] [--no-undistort]\n", argv[0]);
    return -1;
  }
  const std::string in = argv[1], outdir = argv[5];
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), max_frame_counter = std::atoi(argv[4]);
  bool lc = false;
  int levels = 4;
  std::string save_mats, replicate, init_poses;
  bool bgr = false, undistort = true, no_fused = false;
  int world = 1, rank = 0, device = 0, comm_port = 0;
  std::string comm_id_file;
  for (int i = 6; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "LC") lc = true;
    else if (a == "--save-mats" && i + 1 < argc) save_mats = argv[++i];
    else if (a == "--replicate" && i + 1 < argc) replicate = argv[++i];
    else if (a == "--init-poses" && i + 1 < argc) init_poses = argv[++i];
    else if (a == "--bgr") bgr = true;
    else if (a == "--no-undistort") undistort = false;
    else if (a == "--fused") no_fused = false;   // (default) tracked frames through ellc_track_frame (alignment + depth stages as one device sequence)
    else if (a == "--no-fused") no_fused = true;   // every stage as its own call: GetImagePoseEstimate, then observe / regularise / export
    else if (a == "--world" && i + 1 < argc) world = std::atoi(argv[++i]);
    else if (a == "--rank" && i + 1 < argc) rank = std::atoi(argv[++i]);
    else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
    else if (a == "--comm-id" && i + 1 < argc) comm_id_file = argv[++i];
    else if (a == "--comm-tcp" && i + 1 < argc) comm_port = std::atoi(argv[++i]);
    else if (!a.empty() && a[0] >= '0' && a[0] <= '9') levels = std::atoi(a.c_str());
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return -1; }
  }
  const int KEYFRAME_PROPAGATE_INTERVAL = 8;   // ExternVariable.h:39
  std::ifstream f(in, std::ios::binary);
  if (!f) { std::fprintf(stderr, "cannot open %s\n", in.c_str()); return -1; }
  std::ofstream pose_file_orig(outdir + "/poses_orig.txt"), match_file(outdir + "/matchframes.txt");
  if (!pose_file_orig || !match_file) { std::fprintf(stderr, "cannot open output files in %s\n", outdir.c_str()); return -1; }
  try {
    ellc_config cfg;
    ellc_default_config(&cfg, W, H, levels);
    cfg.device = device;   // the tracking context: keyframe slots 0,1 (active / incoming), one alignment at a time; in LC mode the
                           // loop-closure ring and its batches live in a context of their own (facade class globalOptimize)
    Runtime rt(cfg);
    struct CommOwner {   // destroyed after the loop, before the runtime
      ellc_comm* c = nullptr;
      ~CommOwner() { if (c) ellc_comm_destroy(c); }
    } comm;
    if (world > 1) {
      if (rank < 0 || rank >= world || (comm_id_file.empty() && comm_port <= 0)) { std::fprintf(stderr, "--world needs --rank and --comm-id FILE or --comm-tcp PORT\n"); return -1; }
      const int max_total = globalOptimize::MAX_LOOP_ARRAY_LENGTH_SCALE_AVG;
      ellc_status st;
      if (!comm_id_file.empty()) {
        unsigned char id[128];
        if (rank == 0) {
          if (ellc_comm_unique_id(id) != ELLC_OK) { std::fprintf(stderr, "ellc_comm_unique_id failed\n"); return -2; }
          std::remove(comm_id_file.c_str());   // a stale id from an earlier run must not be picked up (give every run a fresh path anyway)
          std::ofstream o((comm_id_file + ".tmp").c_str(), std::ios::binary);
          o.write((const char*)id, 128);
          o.close();
          std::rename((comm_id_file + ".tmp").c_str(), comm_id_file.c_str());
        } else {
          for (int attempt = 0;; attempt++) {
            std::ifstream idf(comm_id_file.c_str(), std::ios::binary);
            if (idf && idf.read((char*)id, 128)) break;
            if (attempt > 600) { std::fprintf(stderr, "no unique id in %s after 60 s\n", comm_id_file.c_str()); return -2; }
            usleep(100000);
          }
        }
        st = ellc_comm_init_rccl(device, id, world, rank, max_total, &comm.c);
      } else {
        st = ellc_comm_init_tcp("127.0.0.1", comm_port, world, rank, max_total, &comm.c);
      }
      if (st != ELLC_OK) { std::fprintf(stderr, "communicator: %s\n", ellc_comm_last_error(comm.c)); return -2; }
      rt.comm = comm.c; rt.world = world; rt.rank = rank;
    }
    rt.FLAG_DO_LOOP_CLOSURE = lc;
    rt.KEYFRAME_PROPAGATE_INTERVAL = KEYFRAME_PROPAGATE_INTERVAL;
    if (!save_mats.empty()) { rt.FLAG_SAVE_MATS = true; rt.SAVED_MATS_PATH = save_mats; }
    if (!replicate.empty()) { rt.FLAG_REPLICATE_POSE_ESTIMATION = true; rt.SAVED_MATS_PATH = replicate; }
    std::ifstream initialize_pose_file;
    if (!init_poses.empty()) {
      initialize_pose_file.open(init_poses.c_str());
      if (!initialize_pose_file) { std::fprintf(stderr, "cannot open %s\n", init_poses.c_str()); return -1; }
    }
    depthMap currentDepthMap(rt);
    std::unique_ptr<globalOptimize> globalOptimizeLoop;
    if (lc) globalOptimizeLoop.reset(new globalOptimize(rt, outdir + "/matchframes_globalopt.txt"));
    std::vector<std::unique_ptr<frame>> frameptr_vector;
    frame* activeKeyFrame = nullptr;
    std::vector<uint8_t> buf(bgr ? (size_t)W * H * 48 : (size_t)W * H);
    if (bgr) {   // the reference's camera is given for 1920 x 1080 (ExternVariable.h:53-62): scale with the input width
      const float sc = (4.0f * W) / 1920.0f;
      const float dist[5] = {-0.288283f, 0.146546f, 0.003800f, -0.001690f, -0.132134f};
      rt.configureIngest(4 * W, 4 * H, 1642.405612f * sc, 1636.148027f * sc, 2.0f * W, 2.0f * H, dist, undistort);
    }
    float initial_pose[6] = {0, 0, 0, 0, 0, 0};
    for (int frame_counter = 1; frame_counter <= max_frame_counter; frame_counter++) {
      if (!f.read((char*)buf.data(), buf.size())) { std::fprintf(stderr, "short read at frame %d\n", frame_counter); return -1; }
      frameptr_vector.emplace_back(new frame(rt, buf.data(), bgr));
      frame* cur = frameptr_vector.back().get();
      if (frame_counter == 1) {   // main.cpp:228-236
        activeKeyFrame = cur;
        currentDepthMap.formDepthMap(cur);
        currentDepthMap.updateDepthImage();
        continue;
      }
      frame* tminus1 = frameptr_vector[frameptr_vector.size() - 2].get();
      const bool have_init = initialize_pose_file.is_open();
      if (have_init) {   // main.cpp:207-211
        int temp_frame_no;
        initialize_pose_file >> temp_frame_no >> initial_pose[0] >> initial_pose[1] >> initial_pose[2] >> initial_pose[3] >> initial_pose[4] >> initial_pose[5];
        if (!initialize_pose_file) { std::fprintf(stderr, "initial-pose file ends before frame %d\n", frame_counter); return -1; }
      }
      // a frame that switches the keyframe (main.cpp:404) is aligned on its own; with --fused every other frame goes through the
      // fused call: alignment + observe + regularise + export as one device sequence (TrackFrameAndObserve)
      const bool kf_switch = (frame_counter % KEYFRAME_PROPAGATE_INTERVAL == 0) || (frame_counter == max_frame_counter);
      float seeds_num = 0;
      if (kf_switch || no_fused) {
        GetImagePoseEstimate(activeKeyFrame, cur, frame_counter, &currentDepthMap, tminus1, have_init ? initial_pose : nullptr);   // main.cpp:330
        seeds_num = currentDepthMap.calculate_no_of_Seeds();
      } else {
        TrackFrameAndObserve(activeKeyFrame, cur, &currentDepthMap, tminus1, have_init ? initial_pose : nullptr, &seeds_num);
      }
      const int id = cur->frameId + rt.BATCH_START_ID - 1, kid = activeKeyFrame->frameId + rt.BATCH_START_ID - 1;
      pose_file_orig << id << " " << kid << " " << cur->poseWrtWorld[0] << " " << cur->poseWrtWorld[1] << " " << cur->poseWrtWorld[2] << " "
                     << cur->poseWrtWorld[3] << " " << cur->poseWrtWorld[4] << " " << cur->poseWrtWorld[5] << " " << activeKeyFrame->rescaleFactor
                     << " " << seeds_num << "\n";
      match_file << id << " " << kid << " " << cur->poseWrtOrigin[0] << " " << cur->poseWrtOrigin[1] << " " << cur->poseWrtOrigin[2] << " "
                 << cur->poseWrtOrigin[3] << " " << cur->poseWrtOrigin[4] << " " << cur->poseWrtOrigin[5] << " " << activeKeyFrame->rescaleFactor
                 << " " << seeds_num << " " << "0" << " " << "0" << " " << "0" << "\n";
      if (kf_switch || no_fused) currentDepthMap.formDepthMap(cur);   // main.cpp:391
      if (kf_switch) {   // main.cpp:404
        if (lc) activeKeyFrame->finaliseWeights();
        currentDepthMap.finaliseKeyframe();
        if (lc) globalOptimizeLoop->pushToArray(activeKeyFrame, &currentDepthMap);   // main.cpp:462
        currentDepthMap.createKeyFrame(cur);
        activeKeyFrame = cur;
        frameptr_vector.erase(frameptr_vector.begin(), frameptr_vector.end() - 1);   // keep only the most recent frame
        continue;
      }
      if (no_fused) {
        currentDepthMap.updateKeyFrame();   // main.cpp:499-502
        currentDepthMap.observeDepthRowParallel();
        currentDepthMap.doRegularization();
        currentDepthMap.updateDepthImage();
      }
    }
    if (globalOptimizeLoop) globalOptimizeLoop->join_all();   // the last keyframe's match thread (t_group.join_all)
  } catch (const std::exception& e) {
    std::fprintf(stderr, "ellc_main: %s\n", e.what());
    return -2;
  }
  return 0;
}
