mkdir -p gpurun_out/d4
R4=$(python -c "print(','.join(['cci']+['cci@%d'%i for i in range(1,4)]))")
R8=$(python -c "print(','.join(['cci']+['cci@%d'%i for i in range(1,8)]))")
R16=$(python -c "print(','.join(['cci']+['cci@%d'%i for i in range(1,16)]))")
R24=$(python -c "print(','.join(['cci']+['cci@%d'%i for i in range(1,24)]))")
E16=$(python -c "print(','.join(['ema']+['ema@%d'%i for i in range(1,16)]))")
E32=$(python -c "print(','.join(['ema']+['ema@%d'%i for i in range(1,32)]))")
T16=$(python -c "print(','.join(['t3']+['t3@%d'%i for i in range(1,16)]))")
T32=$(python -c "print(','.join(['t3']+['t3@%d'%i for i in range(1,32)]))")
timeout -k 10 600 python scripts/exp_time.py cci $R4 $R8 $R16 $R24 ema $E16 $E32 t3 $T16 $T32 volume_all,dm_system_all volume_all,dm_system_all,stochf,ultosc,midprice,macdext,apo_ppo,kama volume_all,dm_system_all,stochf,ultosc,midprice,macdext,apo_ppo,kama,ht_all,stoch > gpurun_out/d4/contention.txt 2>&1
cat gpurun_out/d4/contention.txt
