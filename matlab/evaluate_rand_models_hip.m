function out = evaluate_rand_models_hip( data4sysid_all , varargin )
%evaluate_rand_models_hip: the sweep of evaluate_rand_models.m:45-171 on MI355X GPUs, driven by ONE MATLAB interpreter.
%   out = evaluate_rand_models_hip( data4sysid_all , 'Name' , value , ... )
%   data4sysid_all : the cell array the reference script loads (evaluate_rand_models.m:19-20), one data4sysid struct
%                    (train{:}.t/.y/.u , val{1}.t/.y/.u) per random system.
%   Names          : 'max_degree_linear' (13) , 'max_degree_bilinear' (6) , 'max_degree_nonlinear' (4)   (:14-16)
%                    'lasso_nonlinear' (4)   the L1 parameter of the nonlinear models (:122); linear / bilinear use Inf (:56,:89)
%                    'device_ids' (0)        GPUs to use; more than one: the systems are dealt over them by the library's own
%                                            worker threads (kp_mex('multi_*')) - no parpool, no second MATLAB
%   out            : err_linear_models , err_bilinear_models , err_nonlinear_models   (max_degree x nsystems, :40-42, :76,109,142)
%                    dim_linear_models , dim_bilinear_models , dim_nonlinear_models   (:43-45, :77,110,143)
%                    mean_* , std_* of the kept systems (:149-171) and the percentile bars *_bars (:209-211)
%
%   The reference builds 23 Ksysid objects per system (symbolic dictionary, per-row lift loop, `\` / quadprog, rollout).  Here
%   the raw trials of ALL systems go to the device once (kp_traj_upload: scaling get_scale / scale_data and the snapshot pairs
%   of get_snapshotPairs are formed there), and per model type ONE call serves every degree of every system
%   (kp_sweep_eval_nested: fit + get_model / get_BLmodel / get_NLmodel + val_* rollout + the normalised mean error of :69-72).
%   Systems whose trials do not share one layout, or dictionaries wider than 16 columns, take the loop of the reference with
%   KsysidHip objects instead (same outputs).

    p = inputParser;
    p.addParameter( 'max_degree_linear' , 13 );
    p.addParameter( 'max_degree_bilinear' , 6 );
    p.addParameter( 'max_degree_nonlinear' , 4 );
    p.addParameter( 'lasso_nonlinear' , 4 );
    p.addParameter( 'device_ids' , 0 );
    p.parse( varargin{:} );
    opt = p.Results;
    types = { 'linear' , 'bilinear' , 'nonlinear' };
    maxdeg = [ opt.max_degree_linear , opt.max_degree_bilinear , opt.max_degree_nonlinear ];
    lasso = [ Inf , Inf , opt.lasso_nonlinear ];
    nsys = numel( data4sysid_all );

    [ stacked , Y , U , Yv , Uv , ntrials ] = stack_trials( data4sysid_all );
    n = size( data4sysid_all{1}.train{1}.y , 2 );
    m = size( data4sysid_all{1}.train{1}.u , 2 );
    err = cell( 1 , 3 ); dims = cell( 1 , 3 );
    for k = 1 : 3
        nv = n + m * strcmp( types{k} , 'nonlinear' );
        N = arrayfun( @(d) nchoosek( nv + d , d ) , ( 1 : maxdeg(k) )' );     % functions of the degree-d dictionary (Ksysid.m:641)
        if strcmp( types{k} , 'bilinear' )
            dims{k} = repmat( N * ( m + 1 ) , 1 , nsys );                     % size( basis.full_input , 1 ) (:110)
        else
            dims{k} = repmat( N , 1 , nsys );                                 % size( basis.full , 1 ) (:77, :143)
        end
    end

    widest = [ nchoosek( n + maxdeg(1) , maxdeg(1) ) + m , nchoosek( n + maxdeg(2) , maxdeg(2) ) * ( m + 1 ) , ...
               nchoosek( n + m + maxdeg(3) , maxdeg(3) ) ];
    if stacked && all( widest <= 16 )
        % ---- the device-resident sweep ---------------------------------------------------------------------------------
        multi = numel( opt.device_ids ) > 1;
        if multi
            g = kp_mex( 'multi_create' , opt.device_ids );
            gown = KpOwner( g , 'multi_destroy' ); %#ok<NASGU>
            t = kp_mex( 'multi_traj_upload' , g , Y , U , Yv , Uv , ntrials );
            town = KpOwner( t , 'multi_traj_destroy' ); %#ok<NASGU>
        else
            h = kp_mex( 'create' , opt.device_ids(1) );
            hown = KpOwner( h , 'destroy' );
            t = kp_mex( 'traj_upload' , h , Y , U , Yv , Uv , ntrials );
            town = KpOwner( t , 'traj_destroy' , hown ); %#ok<NASGU>
        end
        for k = 1 : 3
            desc = poly_descriptor( types{k} , n , m , maxdeg(k) );
            las = lasso(k); if isinf( las ), las = 1e6; end                   % Ksysid maps Inf to the least-squares branch (:1068)
            if multi
                [ e , st ] = kp_mex( 'multi_sweep_eval_nested' , g , t , desc , las , maxdeg(k) );
            else
                b = kp_mex( 'basis_create' , h , desc );
                bown = KpOwner( b , 'basis_destroy' , hown );
                [ e , st ] = kp_mex( 'sweep_eval_nested' , h , t , b , las , maxdeg(k) );
                bown.release();
            end
            e = reshape( e( 1 , : , : ) , nsys , maxdeg(k) )';                % first output (:69-72 use the scalar mean of a 1-D system)
            e( st' ~= 0 ) = NaN;                                              % a system whose fit failed has no model
            err{k} = e;
        end
    else
        % ---- systems of differing layouts / wide dictionaries: the reference's loops with KsysidHip (:47-144) ------------
        for k = 1 : 3
            err{k} = zeros( maxdeg(k) , nsys );
        end
        for i = 1 : nsys
            for k = 1 : 3
                for j = 1 : maxdeg(k)
                    sysid = KsysidHip( data4sysid_all{i} , 'model_type' , types{k} , 'time_type' , 'discrete' , 'obs_type' , { 'poly' } , ...
                                       'obs_degree' , j , 'snapshots' , Inf , 'lasso' , lasso(k) , 'delays' , 0 , 'loaded' , false , ...
                                       'dim_red' , false );
                    sysid.koopData_PxPy = false;
                    sysid = sysid.train_models;
                    results = sysid.valNplot_model( [] , false , false );
                    mean_error_zeros = sum( abs( results{1}.real.y ) ) / size( results{1}.real.y , 1 );    % :70
                    e = results{1}.error.mean / mean_error_zeros;                                          % :72
                    err{k}( j , i ) = e(1);
                    sysid.delete_hip();
                end
            end
        end
    end

    out.err_linear_models = err{1}; out.err_bilinear_models = err{2}; out.err_nonlinear_models = err{3};
    out.dim_linear_models = dims{1}; out.dim_bilinear_models = dims{2}; out.dim_nonlinear_models = dims{3};
    % ---- statistics (:149-171, :209-211): a system is kept when ALL its degrees are below 10 (NaN fails the comparison) ----
    names = { 'linear' , 'bilinear' , 'nonlinear' };
    for k = 1 : 3
        kept = err{k}( : , all( err{k} < 10 ) );
        out.( [ 'mean_' , names{k} ] ) = mean( kept , 2 );
        out.( [ 'std_' , names{k} ] ) = std( kept' )';
        out.( [ names{k} , '_bars' ] ) = prctile( kept , [ 0 25 50 75 100 ] , 2 );
    end
end

function desc = poly_descriptor( model_type , n , m , deg )
    % the dictionary of KsysidHip.hip_descriptor for obs_type {'poly'}, obs_degree deg, no dim_red (def_polyLift, Ksysid.m:645-648)
    nv = n + m * strcmp( model_type , 'nonlinear' );
    e = [];
    for d = 1 : deg
        e = [ e ; partitions( d , ones( 1 , nv ) ) ]; %#ok<AGROW>
    end
    e = e( nv + 1 : end , : );
    mt = find( strcmp( model_type , { 'linear' , 'bilinear' , 'nonlinear' } ) ) - 1;
    desc = struct( 'model_type' , int32( mt ) , 'nzeta' , int32( n ) , 'm' , int32( m ) , 'block_type' , int32( 0 ) , ...
                   'block_count' , int32( size( e , 1 ) ) , 'poly_exps' , uint8( e' ) , 'gauss_centres' , [] , 'pcs' , [] );
end

function [ ok , Y , U , Yv , Uv , k ] = stack_trials( data4sysid_all )
    % merged training trials (Ksysid.merge_trials, :380-401) of every system as rows x n x nsys blocks, RAW values; ok = false
    % unless every system has the same number of equally long trials whose clocks restart at every join (get_snapshotPairs
    % drops the pair across a join when before.t >= after.t, :948 - the device forms the pairs under that rule)
    nsys = numel( data4sysid_all );
    k = numel( data4sysid_all{1}.train );
    T = size( data4sysid_all{1}.train{1}.y , 1 );
    Tv = size( data4sysid_all{1}.val{1}.y , 1 );
    n = size( data4sysid_all{1}.train{1}.y , 2 );
    m = size( data4sysid_all{1}.train{1}.u , 2 );
    Y = zeros( k * T , n , nsys ); U = zeros( k * T , m , nsys ); Yv = zeros( Tv , n , nsys ); Uv = zeros( Tv , m , nsys );
    ok = true;
    for i = 1 : nsys
        d = data4sysid_all{i};
        if numel( d.train ) ~= k || ~isequal( size( d.val{1}.y ) , [ Tv , n ] ) || ~isequal( size( d.val{1}.u ) , [ Tv , m ] )
            ok = false; return;
        end
        tprev = -Inf;
        for j = 1 : k
            tr = d.train{j};
            if ~isequal( size( tr.y ) , [ T , n ] ) || ~isequal( size( tr.u ) , [ T , m ] ) || any( diff( tr.t(:) ) <= 0 ) || ...
                    ( j > 1 && tprev < tr.t(1) )
                ok = false; return;
            end
            tprev = tr.t(end);
            Y( (j-1)*T+1 : j*T , : , i ) = tr.y;
            U( (j-1)*T+1 : j*T , : , i ) = tr.u;
        end
        Yv( : , : , i ) = d.val{1}.y;
        Uv( : , : , i ) = d.val{1}.u;
    end
end
