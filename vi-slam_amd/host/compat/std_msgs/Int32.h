// compat/std_msgs/Int32.h -- EMPTY stand-in: /root/reference/src/main_vi_slamGPU.cpp:13-18 includes <std_msgs/Int32.h> but calls nothing from it (its ROS users,
// src/Visualizer.cpp and the ROS node mains, are out of scope: SURVEY 8 / DESIGN 6).  Lets the reference's file compile unchanged where ROS is absent.
